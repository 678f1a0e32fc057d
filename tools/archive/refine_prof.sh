#!/bin/bash
# usage (on the GPU box): tools/refine_prof.sh   -- kernel trace of Gauss-Newton updates of the 10 000-pose window (refined, partitioned form)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_refine
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/tools/refine_probe.py gnp > $OUT/run.log 2> $OUT/run.err
python3 - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
print("kernel,calls,total_ms,avg_us")
for r in rows[:22]:
    print(f'{r["Name"][:60]},{r["Calls"]},{float(r["TotalDurationNs"])/1e6:.3f},{float(r["AverageNs"])/1e3:.1f}')
PY
