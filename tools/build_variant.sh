#!/bin/bash
# usage: tools/build_variant.sh <name> [-DFLAG ...]   -> tools/variants/libvilfusion_<name>.so
# Builds vf_kernels.hip with extra defines against the current objects of the other translation
# units; used to A/B tuning parameters on the GPU box (copy the variant over libvilfusion.so).
set -e
name=$1; shift
R=$(cd "$(dirname "$0")/.." && pwd)
C=$R/vil_sensor_fusion_amd/csrc
make -C $C -s -j4
mkdir -p $R/tools/variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -Wall -Wno-unused-function "$@" \
    -c $C/vf_kernels.hip -o $R/tools/variants/vf_kernels_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/variants/libvilfusion_$name.so \
    $R/tools/variants/vf_kernels_$name.o $C/build/vf_engine.o $C/build/vf_degeneracy.o $C/build/vf_refine.o $C/build/vf_graph.o
rm -f $R/tools/variants/vf_kernels_$name.o
echo built $name
