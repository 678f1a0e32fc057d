#!/bin/bash
# usage (GPU box, repo root): tools/ab_variants.sh <name> ...   bench line summary per variant under tools/variants/
R=$GRAFT_REPO_ROOT
cp $R/vil_sensor_fusion_amd/libvilfusion.so /tmp/libvf_orig.so
for n in "$@"; do
  cp $R/tools/variants/libvilfusion_$n.so $R/vil_sensor_fusion_amd/libvilfusion.so
  python3 -m pytest $R/tests/test_gpu_parity.py $R/tests/test_gpu_edge_cases.py -q -x -m gpu > $R/gpurun_out/ab_$n.test 2>&1; echo "$n tests rc=$? $(tail -1 $R/gpurun_out/ab_$n.test)"
  python3 $R/bench.py --steps 6 --warmup 2 --no-sharded --no-cpu-baseline --no-convergence-exit --no-degeneracy --no-graph-manager 2> $R/gpurun_out/ab_$n.err | tail -1 > $R/gpurun_out/ab_$n.json
  python3 -c "
import json,sys
d=json.load(open('$R/gpurun_out/ab_$n.json'))
print('$n', round(d['value']), 'kf/s', round(d['ms_per_step'],2), 'ms/step', {k: round(v,3) for k,v in d['stage_ms'].items()}, 'single', round(d['single_window']['ms_per_update'],3), round(d['single_window']['solve_ms'],4))
"
done
cp /tmp/libvf_orig.so $R/vil_sensor_fusion_amd/libvilfusion.so
