"""Kernel boundaries of a graph-replayed train step, from a rocprofv3 kernel trace: for every kernel of the steady-state steps
the GAP between its start and the latest end of any kernel that started before it (0 when it overlaps a predecessor: side-stream
branches), i.e. the time the chip is idle at a dependent boundary -- NOT the duration of the smallest kernels, which is what
DESIGN.md quoted as "launch floor" until round 3.

usage: python tools/trace_gaps.py <p_kernel_trace.csv> [steps_to_skip_at_the_head] > profiles/rNN_x_gaps.txt
The step boundary is the plan's fill_zero_multi_kernel launch (first kernel of every step)."""
import csv
import statistics
import sys


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "")
    return name.split("(")[0][:70]


def main():
    path = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in csv.DictReader(open(path))]
    rows.sort()
    starts = [i for i, r in enumerate(rows) if r[2].startswith("fill_zero_multi_kernel")]
    if len(starts) < skip + 3:
        print("too few steps in the trace (%d)" % len(starts))
        return
    steps = [(starts[i], starts[i + 1]) for i in range(skip, len(starts) - 1)]
    gaps, by_pred, busy, wall, per_step_gap = [], {}, [], [], []
    for (a, b) in steps:
        last_end, last_name = rows[a][1], rows[a][2]
        g_sum = 0.0
        t_busy = rows[a][1] - rows[a][0]
        for i in range(a + 1, b):
            s, e, n = rows[i]
            gap = max(0, s - last_end)
            if s >= last_end:                        # a dependent boundary on an otherwise idle chip
                gaps.append(gap / 1e3)
                by_pred.setdefault(last_name, []).append(gap / 1e3)
                g_sum += gap / 1e3
                t_busy += e - s
            else:
                t_busy += max(0, e - max(s, last_end))
            if e > last_end:
                last_end, last_name = e, n
        per_step_gap.append(g_sum)
        busy.append(t_busy / 1e3)
        wall.append((last_end - rows[a][0]) / 1e3)
    q = lambda xs, p: sorted(xs)[min(len(xs) - 1, int(p * len(xs)))]
    print("steps analysed: %d (of %d in the trace), kernels per step: %d" % (len(steps), len(starts), steps[0][1] - steps[0][0]))
    print("step wall (first kernel start -> last kernel end): median %.1f us; chip busy (union of kernel intervals): %.1f us; idle at boundaries: %.1f us" % (
        statistics.median(wall), statistics.median(busy), statistics.median(per_step_gap)))
    print("dependent boundaries per step: %.0f; gap (successor start - predecessor end): median %.2f us, p10 %.2f, p90 %.2f, max %.1f" % (
        len(gaps) / len(steps), statistics.median(gaps), q(gaps, 0.1), q(gaps, 0.9), max(gaps)))
    print("\nlargest total idle time by predecessor kernel (us per step, boundaries per step, median gap):")
    tot = sorted(((sum(v) / len(steps), len(v) / len(steps), statistics.median(v), k) for k, v in by_pred.items()), reverse=True)
    for t, n, med, k in tot[:25]:
        print("  %7.2f  %5.1f  %6.2f  %s" % (t, n, med, k))


if __name__ == "__main__":
    main()
