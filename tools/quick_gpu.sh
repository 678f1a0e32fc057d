#!/bin/bash
# usage (GPU box, repo root): tools/quick_gpu.sh <tag>  -> GPU tests + the short bench summary (+ K4 stamps if the stamps library exists)
R=$GRAFT_REPO_ROOT; tag=$1
python3 -m pytest $R/tests -m gpu -x -q > $R/gpurun_out/gpu_tests_$tag.log 2>&1; echo "tests rc=$? $(tail -1 $R/gpurun_out/gpu_tests_$tag.log)"
python3 $R/bench.py --no-cpu-baseline --no-degeneracy 2> $R/gpurun_out/bench_$tag.err | tail -1 > $R/gpurun_out/bench_$tag.json
python3 -c "
import json
d=json.load(open('$R/gpurun_out/bench_$tag.json'))
print(round(d['value']), 'kf/s', round(d['ms_per_step'],2), 'ms/step', {k: round(v,3) for k,v in d['stage_ms'].items()}, 'single', round(d['single_window']['ms_per_update'],3), round(d['single_window']['solve_ms'],4), 'sharded', round(d['time_sharded_window']['ms_per_lm_trial'],3), 'gm', round(d['graph_manager']['solve_ms_mean'],3), 'conv', round(d['with_convergence_exit']['value']))
"
