#!/bin/bash
# usage: run_with_lib.sh <variant.so> <command...>: run a command with the library replaced by a variant
cp vil_sensor_fusion_amd/libvilfusion.so /tmp/lib_backup.so
cp $1 vil_sensor_fusion_amd/libvilfusion.so; shift
"$@"
cp /tmp/lib_backup.so vil_sensor_fusion_amd/libvilfusion.so
