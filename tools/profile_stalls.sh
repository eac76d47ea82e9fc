#!/bin/bash
# Stall breakdown of one eager bench run, per kernel instantiation: two --pmc passes (kernel trace only beside them)
#   SQ pass : SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES
#   TCP pass: TCP_PENDING_STALL_CYCLES TCP_GATE_EN1 (TCP_COUNTERS overrides)
# -> profiles/<tag>_pmc_stalls.txt       usage (GPU box, repo root): [BENCH_ARGS=...] bash tools/profile_stalls.sh r04_a
set -e -o pipefail
TAG=${1:-r04_x}
OUT=gpurun_out/stalls_$TAG
export TMPDIR=/tmp
mkdir -p $OUT
PCMD="bench.py --steps 3 --warmup 1 --windows 1 --no-graphs --profile-steps 0 --no-cpu-baseline --no-segmented --no-other-configs $BENCH_ARGS"
rocprofv3 -L > $OUT/counters_list.txt 2>&1 || true
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES \
  -d $OUT/sq -o p --output-format csv -- python3 $PCMD > $OUT/bench_sq.json 2> $OUT/sq.err
echo "SQ stall pass done"
# (five TCP counters in one pass: "Request exceeds the capabilities of the hardware to collect", and rocprofv3 then hangs in its
# signal handler -- hence two counters and a timeout)
TCPC="${TCP_COUNTERS:-TCP_PENDING_STALL_CYCLES TCP_GATE_EN1}"
echo "TCP counters: $TCPC"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc $TCPC -d $OUT/tcp -o p --output-format csv -- python3 $PCMD > $OUT/bench_tcp.json 2> $OUT/tcp.err || echo "TCP pass failed (see $OUT/tcp.err)"
python3 profiles/stall_table.py $(find $OUT/sq -name '*counter_collection.csv' | head -1) "$(find $OUT/tcp -name '*counter_collection.csv' | head -1)" "python3 $PCMD" > profiles/${TAG}_pmc_stalls.txt
cp profiles/${TAG}_pmc_stalls.txt $OUT/
head -40 profiles/${TAG}_pmc_stalls.txt
