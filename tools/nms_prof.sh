#!/bin/bash
# kernel-only durations of the NMS kernels over tools/nms_bench.py's geometries (rocprofv3 kernel trace)
export TMPDIR=/tmp
OUT=gpurun_out/prof_nms_$$
rocprofv3 --kernel-trace -d $OUT -o p --output-format csv -- python3 tools/nms_bench.py > /dev/null 2>&1
python3 - <<PYEOF
import csv
rows = list(csv.DictReader(open("$OUT/p_kernel_trace.csv")))
seq = [(r["Kernel_Name"][:40], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r["Grid_Size_X"]) for r in rows if "nms" in r["Kernel_Name"]]
names = ["rpn 8768->300 uniform", "rpn 8768->8", "rpn 8768->300 clustered", "rpn 8768->300 equal", "rpn 22464->300", "det 300x7", "det 300x7->8", "det 300x7 clustered"]
for n, i in zip(names, range(0, len(seq), 46)):
    blk = seq[i:i+46]
    cls = sorted(x[1] for x in blk if "class" in x[0]); mrg = sorted(x[1] for x in blk if "merge" in x[0])
    print("%-26s class %.1f us  merge %.1f us" % (n, cls[len(cls)//2], mrg[len(mrg)//2]))
PYEOF
