#!/bin/bash
# diagnostic build of the library with s_memtime stamps in the band solver (never shipped)
set -e
cd "$(dirname "$0")/../vil_sensor_fusion_amd/csrc"
mkdir -p build_stamps
for f in vf_kernels vf_engine vf_degeneracy vf_refine; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -DVF_SOLVE_STAMPS -c $f.hip -o build_stamps/$f.o &
done
wait
g++ -O2 -std=c++17 -fPIC -c vf_graph.cpp -o build_stamps/vf_graph.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../tools/libvilfusion_stamps.so build_stamps/*.o
