#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>
tag=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
FLAGS="--host-workers 1 --steps 10 --warmup 2 --no-single-window --no-sharded --no-cpu-baseline --no-convergence-exit --no-degeneracy --no-graph-manager"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/bench.py $FLAGS > $OUT/bench_traced.json 2> $OUT/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $R/bench.py --host-workers 1 --steps 3 --warmup 1 --no-single-window --no-sharded --no-cpu-baseline --no-convergence-exit --no-degeneracy --no-graph-manager > /dev/null 2> $OUT/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 $R/bench.py --host-workers 1 --steps 3 --warmup 1 --no-single-window --no-sharded --no-cpu-baseline --no-convergence-exit --no-degeneracy --no-graph-manager > /dev/null 2> $OUT/write.err
cd $R
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
ls -R $OUT | head -30
