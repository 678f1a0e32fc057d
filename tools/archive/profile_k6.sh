#!/bin/bash
# usage (GPU box, repo root): tools/profile_k6.sh <tag>  -> gpurun_out/k6_<tag>/k6_pmc.md (+ the probe's own timing)
tag=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/k6_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $R/tools/k6_probe.py > $OUT/k6_probe.log 2>&1
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 $R/tools/k6_probe.py > /dev/null 2> $OUT/trace.err
echo trace done >> $OUT/k6_probe.log
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/pmc -o p -- python3 $R/tools/k6_probe.py > /dev/null 2> $OUT/pmc.err
echo pmc done >> $OUT/k6_probe.log
python3 $R/tools/k6_counters.py $(find $OUT/pmc -name "*counter_collection.csv" | head -1) $(find $OUT/trace -name "*kernel_trace.csv" | head -1) $OUT/k6_pmc.md > $OUT/summary.log 2>&1
rm -rf $OUT/trace $OUT/pmc
cat $OUT/summary.log
