"""One replayed training step as a timeline, from a rocprofv3 kernel trace (tools/prof_mode.sh keeps <tag>_<i>_kernel_trace.csv):
start (us from the step's first kernel), duration, hardware queue, `||` where the kernel starts before its predecessor has ended (a
side-stream branch), kernel name.  The step boundary is the plan's first fill_zero_multi_kernel launch.
usage: python tools/step_timeline.py <kernel_trace.csv> [step index, default 20] > profiles/rNN_step_timeline.txt"""
import csv
import re
import sys


def main():
    path = sys.argv[1]
    which = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", ""), int(r["Queue_Id"]))
            for r in csv.DictReader(open(path))]
    rows.sort()
    fills = [i for i, r in enumerate(rows) if r[2].startswith("fill_zero_multi_kernel")]
    # (the backward pass's accumulation targets have a fill of their own on a side stream, a few us after the step's first one: a fill that
    # starts within 300 us of the previous one does not open a step)
    starts = [i for n, i in enumerate(fills) if n == 0 or rows[i][0] - rows[fills[n - 1]][0] > 300_000]
    a, b = starts[which], starts[which + 1]
    t0 = rows[a][0]
    prev_end, busy = t0, 0
    print("# step %d of %s: %d kernels, %.1f us from the first kernel's start to the next step's first kernel" % (which, path, b - a, (rows[b][0] - t0) / 1e3))
    print("# start_us  dur_us  queue  kernel  (|| = starts before its predecessor has ended)")
    for i in range(a, b):
        s, e, n, q = rows[i]
        gap = (s - prev_end) / 1e3
        name = re.sub(r"\(.*", "", n)
        name = re.sub(r"^_ZN12_GLOBAL__N_1\d+", "", name)[:100]
        print("%9.1f %7.1f  q%d %s %s%s" % ((s - t0) / 1e3, (e - s) / 1e3, q, "  " if s >= prev_end - 50 else "||", name,
                                           "   <- %.1f us idle" % gap if gap >= 4.0 else ""))
        prev_end = max(prev_end, e)


if __name__ == "__main__":
    main()
