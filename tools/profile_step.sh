#!/bin/bash
# One bench command under rocprofv3, three separate passes (kernel trace + stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE -- the two
# TCC counters do not fit one pass, and gpurun refuses PMC together with trace domains other than the kernel trace), reduced
# into the tracked files under profiles/:  <tag>_kernel_stats.csv, <tag>_summary.txt, <tag>_hbm_per_kernel.csv, rNN_offline$OFFLINE_VARIANT.json
# usage (on the GPU box, from the repo root):  [BENCH_ARGS="--fp8 ..." OFFLINE_VARIANT=_fp8] bash tools/profile_step.sh <tag>
set -e -o pipefail
TAG=${1:-r02_x}
OUT=gpurun_out/prof_$TAG
# 1 eager warm-up run (plan build) + 4 warm-up steps + 25 timed steps = 30 executions of the step's kernels
CMD="bench.py --steps 25 --warmup 4 --windows 1 --profile-steps 0 --no-cpu-baseline --no-segmented --no-other-configs $BENCH_ARGS"
STEPS=30
export TMPDIR=/tmp
mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/trace -o p --output-format csv -- python3 $CMD > $OUT/bench_trace.json 2> $OUT/trace.err
echo "trace pass done"
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/fetch -o p --output-format csv -- python3 $CMD > $OUT/bench_fetch.json 2> $OUT/fetch.err
echo "FETCH_SIZE pass done"
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $OUT/write -o p --output-format csv -- python3 $CMD > $OUT/bench_write.json 2> $OUT/write.err
echo "WRITE_SIZE pass done"
STATS=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
FETCH=$(find $OUT/fetch -name '*counter_collection.csv' | head -1)
WRITE=$(find $OUT/write -name '*counter_collection.csv' | head -1)
cp $STATS profiles/${TAG}_kernel_stats.csv
python3 profiles/summarize.py $STATS $STEPS 60 > profiles/${TAG}_summary.txt
python3 profiles/hbm_traffic.py $TAG $STEPS $STATS $FETCH $WRITE "rocprofv3 --kernel-trace --stats | --pmc FETCH_SIZE | --pmc WRITE_SIZE -- python3 $CMD" > $OUT/families.json
cp profiles/${TAG}_kernel_stats.csv profiles/${TAG}_summary.txt profiles/${TAG}_hbm_per_kernel.csv profiles/${TAG:0:3}_offline$OFFLINE_VARIANT.json $OUT/
head -12 profiles/${TAG}_summary.txt
# optional 4th pass (PMC_TABLE=1): SQ counters of an eager run -> profiles/<tag>_pmc_counters.txt (matrix-pipe / LDS busy per kernel)
if [ -n "$PMC_TABLE" ]; then
  PCMD="bench.py --steps 3 --warmup 1 --windows 1 --no-graphs --profile-steps 0 --no-cpu-baseline --no-segmented --no-other-configs $BENCH_ARGS"
  rocprofv3 --kernel-trace --pmc SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU -d $OUT/sq -o p --output-format csv -- python3 $PCMD > $OUT/bench_sq.json 2> $OUT/sq.err
  python3 profiles/pmc_table.py $(find $OUT/sq -name '*counter_collection.csv' | head -1) "python3 $PCMD" > profiles/${TAG}_pmc_counters.txt
  cp profiles/${TAG}_pmc_counters.txt $OUT/
  echo "SQ counter pass done"
fi
