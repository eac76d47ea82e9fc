#!/bin/bash
# A/B of kernel variants inside the sweep library on ONE box: every entry of AB_LIST is NAME=VALUE exported for one bench run.
# usage (GPU box):  AB_LIST="FRCNN_BN_VAR=0 FRCNN_BN_VAR=4 FRCNN_BN_VAR=0" bash tools/ab_lib.sh
i=0
for v in ${AB_LIST}; do
  i=$((i+1))
  env FRCNN_LIB=lib2dod_hip_sweep.so $v timeout -k 10 200 python bench.py --no-cpu-baseline --profile-steps 0 --windows 3 > gpurun_out/abl_$i.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/abl_$i.json"))
print("%-30s" % "$v", d["windows"]["ms_per_step"])
PY
done
