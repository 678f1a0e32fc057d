#!/bin/bash
# usage: run_variant_tests.sh <variant.so>  -- run solve-related GPU tests against a library variant
cp vil_sensor_fusion_amd/libvilfusion.so /tmp/lib_backup.so
cp $1 vil_sensor_fusion_amd/libvilfusion.so
python -m pytest tests/test_gpu_parity.py tests/test_gpu_partitioned.py -x -q -s 2>&1 | grep -i "backward error\|passed\|failed" | tail -12
cp /tmp/lib_backup.so vil_sensor_fusion_amd/libvilfusion.so
