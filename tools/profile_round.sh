#!/bin/bash
# usage (on the GPU box, from the repo root): tools/profile_round.sh <tag>
# Kernel trace + stats of the default bench command, then hardware counters in SEPARATE rocprofv3 passes (a --pmc pass never
# carries a trace domain; the program itself follows `--`): HBM traffic (FETCH_SIZE / WRITE_SIZE, one pass each: the TCC
# block has 4 slots and they cost 3 + 2) and two passes of SQ counters (8 slots each) for MFMA utilisation, VALU issue
# occupancy, LDS bank conflicts and the wait / issue-stall split.  tools/summarize_prof.py condenses them into profiles/.
tag=$1
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$tag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
QUIET="--host-workers 1 --no-single-window --no-sharded --no-accuracy --no-convergence-exit --no-incremental --no-degeneracy --no-graph-manager"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o t -- python3 $R/bench.py $QUIET --steps 10 --warmup 2 > $OUT/bench_traced.json 2> $OUT/trace.err
echo "trace done" > $OUT/progress.txt
PMCRUN="$QUIET --steps 3 --warmup 1 --init-iterations 20"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $R/bench.py $PMCRUN > /dev/null 2> $OUT/fetch.err
echo "fetch done" >> $OUT/progress.txt
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 $R/bench.py $PMCRUN > /dev/null 2> $OUT/write.err
echo "write done" >> $OUT/progress.txt
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq_a -o a -- python3 $R/bench.py $PMCRUN > /dev/null 2> $OUT/sq_a.err
echo "sq_a done" >> $OUT/progress.txt
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_WAVE_CYCLES --output-format csv -d $OUT/sq_b -o b -- python3 $R/bench.py $PMCRUN > /dev/null 2> $OUT/sq_b.err
echo "sq_b done" >> $OUT/progress.txt
rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CU_CYCLES --output-format csv -d $OUT/sq_c -o c -- python3 $R/bench.py $PMCRUN > /dev/null 2> $OUT/sq_c.err
echo "sq_c done" >> $OUT/progress.txt
cd $R
python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
echo "bench done" >> $OUT/progress.txt
# summarise on the box (the per-dispatch counter CSVs of six passes exceed what gpurun carries back), keep the summaries, the
# rocprofv3 --stats CSV and the bench lines
python3 tools/summarize_prof.py $tag $OUT $OUT/summary > $OUT/summarize.log 2>&1
for d in trace fetch write sq_a sq_b sq_c; do rm -rf $OUT/$d; done
du -sh $OUT
