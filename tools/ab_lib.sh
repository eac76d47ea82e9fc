#!/bin/bash
# A/B of kernel variants on ONE box (box-to-box variance is larger than most effects): every entry of AB_LIST is a list of
# NAME=VALUE pairs joined by "+" and exported for one bench run; FRCNN_LIB selects the library (default: the sweep library).
# usage (GPU box):  AB_LIST="FRCNN_BN_VAR=0 FRCNN_BN_VAR=4 FRCNN_BN_VAR=0" bash tools/ab_lib.sh
#                   AB_LIST="FRCNN_LIB=lib2dod_hip_old.so FRCNN_LIB=lib2dod_hip_new.so" bash tools/ab_lib.sh
i=0
for v in ${AB_LIST}; do
  i=$((i+1))
  env FRCNN_LIB=lib2dod_hip_sweep.so ${v//+/ } timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-configs --no-segmented --profile-steps 0 --windows 3 $BENCH_ARGS > gpurun_out/abl_$i.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/abl_$i.json"))
print("%-40s" % "$v", d["windows"]["ms_per_step"])
PY
done
