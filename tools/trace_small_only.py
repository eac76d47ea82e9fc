"""Time per step during which ONLY one-workgroup-per-image kernels (NMS, target assignment, sampling, losses, head post / grad) are
running -- the chip is then all but idle.  usage: python tools/trace_small_only.py <rocprofv3 kernel_trace.csv> [steps]"""
import collections
import csv
import re
import sys

SMALL = ("assign_targets", "sample_kernel", "nms_class", "nms_merge", "losses_kernel", "roi_levels", "rpn_head_grad", "rpn_head_post", "decode",
         "step_inc", "fp8_update_scales")


def short(n):
    m = re.search(r"(\w+_kernel)", n)
    return m.group(1) if m else n[:30]


def main():
    rows = []
    with open(sys.argv[1]) as f:
        for x in csv.DictReader(f):
            rows.append((int(x["Start_Timestamp"]), int(x["End_Timestamp"]), x["Kernel_Name"]))
    rows.sort()
    nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    starts = [s for s, e, n in rows if "fill_zero_multi" in n]
    res = collections.Counter()
    wall = 0.0
    for i in range(len(starts) - nsteps - 1, len(starts) - 1):
        a, b = starts[i], starts[i + 1]
        wall += (b - a) / 1e3
        ev = []
        for s, e, n in rows:
            if a <= s < b:
                ev.append((s, 1, n))
                ev.append((e, -1, n))
        ev.sort()
        active = collections.Counter()
        last = None
        for t, d, n in ev:
            if last is not None and t > last and sum(active.values()) > 0:
                names = [k for k, v in active.items() if v > 0]
                if all(any(sm in k for sm in SMALL) for k in names):
                    res[" + ".join(sorted(set(short(k) for k in names)))] += (t - last) / 1e3
            active[n] += d
            last = t
    print("steps analysed: %d, step wall %.1f us" % (nsteps, wall / nsteps))
    for k, v in res.most_common(12):
        print("%8.1f us/step  %s" % (v / nsteps, k))
    print("total time per step with only one-workgroup-per-image kernels running: %.1f us" % (sum(res.values()) / nsteps))


if __name__ == "__main__":
    main()
