#!/bin/bash
# kernel-trace summary of one bench command per environment variant (same box): usage  MODES="FRCNN_BN_IN=3x3 FRCNN_BN_IN=1" bash tools/prof_mode.sh <tag>
# writes gpurun_out/<tag>_<i>_summary.txt (profiles/summarize.py of the kernel stats, 30 step executions)
set -o pipefail
TAG=${1:-pm}
export TMPDIR=/tmp
i=0
for v in $MODES; do
  i=$((i+1))
  OUT=gpurun_out/prof_${TAG}_$i
  mkdir -p $OUT
  export ${v//+/ }
  rocprofv3 --kernel-trace --stats -d $OUT/trace -o p --output-format csv -- python3 bench.py --steps 25 --warmup 4 --windows 1 --profile-steps 0 --no-cpu-baseline --no-segmented --no-other-configs $BENCH_ARGS > $OUT/bench.json 2> $OUT/trace.err
  STATS=$(find $OUT/trace -name '*kernel_stats.csv' | head -1)
  python3 profiles/summarize.py $STATS 30 90 150 > gpurun_out/${TAG}_${i}_summary.txt
  echo "== $v"; head -3 gpurun_out/${TAG}_${i}_summary.txt
  cp $(find $OUT/trace -name '*kernel_trace.csv' | head -1) gpurun_out/${TAG}_${i}_kernel_trace.csv
  rm -rf $OUT/trace
done
