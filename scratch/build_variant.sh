#!/bin/bash
set -e
name=$1; shift
cd "$(dirname "$0")/../vil_sensor_fusion_amd/csrc"
mkdir -p build_var_$name
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast "$@" -c vf_kernels.hip -o build_var_$name/vf_kernels.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../scratch/libvf_$name.so build_var_$name/vf_kernels.o build/vf_engine.o build/vf_degeneracy.o build/vf_graph.o
