#!/bin/bash
# usage (GPU box, repo root): tools/ab_asm_variants.sh <name> ...   parity tests + stage time of the assembling sweep per library variant (tools/variants/)
R=$GRAFT_REPO_ROOT
cp $R/vil_sensor_fusion_amd/libvilfusion.so /tmp/libvf_orig.so
for n in "$@"; do
  cp $R/tools/variants/libvilfusion_$n.so $R/vil_sensor_fusion_amd/libvilfusion.so
  python3 -m pytest $R/tests/test_gpu_assembling_sweep.py -q -x -m gpu > $R/gpurun_out/ab_$n.test 2>&1; echo "$n tests rc=$? $(tail -1 $R/gpurun_out/ab_$n.test)"
  python3 $R/tools/asm_probe.py 1024 2>&1 | grep "assembling"
done
cp /tmp/libvf_orig.so $R/vil_sensor_fusion_amd/libvilfusion.so
