#!/bin/bash
# What does each side branch of the train plan buy?  A/B on ONE box (box-to-box variance is larger than most of the effects):
# every entry of AB_LIST is a comma list of branch names kept on the main stream (FRCNN_NOBRANCH), "-" = the plan as built,
# "serial" = no branch at all.   usage (GPU box):  AB_LIST="- rpn_side detections rcnn_targets serial -" bash tools/ab_plan.sh
i=0
for v in ${AB_LIST:-- rpn_side detections rcnn_targets serial -}; do
  i=$((i+1))
  nb=$v; ser=""
  [ "$v" = "-" ] && nb=""
  [ "$v" = "serial" ] && nb="" && ser=1
  FRCNN_SERIAL_PLAN=$ser FRCNN_NOBRANCH=$nb timeout -k 10 200 python bench.py --no-cpu-baseline --profile-steps 0 --windows 3 > gpurun_out/ab_$i.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$i.json"))
print("%-40s" % "$v", d["windows"]["ms_per_step"])
PY
done
