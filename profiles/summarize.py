"""Summarise a rocprofv3 kernel_stats.csv: python profiles/summarize.py <csv> [steps]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total GPU kernel time %.3f ms (%.3f ms/step over %g steps)" % (tot / 1e6, tot / 1e6 / steps, steps))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 28]:
    n = r["Name"].replace("(anonymous namespace)::", "")
    w = int(sys.argv[4]) if len(sys.argv) > 4 else 64          # name width (the conv kernels' trailing template arguments need ~130)
    print("%-*s calls/step %7.1f  ms/step %8.3f  avg %8.2f us  %5.1f%%" % (
        w, n[:w], float(r["Calls"]) / steps, float(r["TotalDurationNs"]) / 1e6 / steps, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
