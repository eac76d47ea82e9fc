"""Reduce the rocprofv3 outputs of ONE bench command into the tracked files bench.py and DESIGN.md cite:

  profiles/<tag>_hbm_per_kernel.csv   per kernel: launches, FETCH_SIZE (raw KiB and corrected bytes), WRITE_SIZE, bytes / launch
  profiles/rNN_offline<variant>.json  (rNN = the tag's round prefix, variant = $OFFLINE_VARIANT, e.g. "_fp8") per kernel family: launches / step, average launch duration (kernel trace), HBM bytes per
                                      launch (PMC), stamped with the hash of the kernel sources they were measured on

Counters are collected and corrected as /opt/skills/guides/MI355X_MICROARCH.md 'HBM' prescribes for gfx950: FETCH_SIZE and
WRITE_SIZE in SEPARATE --pmc passes (they do not fit one pass), FETCH_SIZE counts the 128-byte requests of wide coalesced
reads at 64 B -> doubled; WRITE_SIZE is exact for 16-B/lane stores and float atomics.  Both are reported in KiB per dispatch.

usage: python profiles/hbm_traffic.py <tag> <steps> <kernel_stats.csv> <fetch_counter_collection.csv> <write_counter_collection.csv> "<command>"
"""
import collections
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def family(name):
    if "conv_stem_kernel" in name or "conv1x1_stream_kernel" in name or "conv3x3_patch_kernel" in name or "conv3x3_wres_kernel" in name:      # (round 4: the stem's own kernel, the streaming 1x1 kernel -- bf16)
        return "conv fprop/dgrad (conv_tile_kernel)"
    if "conv_tile_kernel" in name:
        # the last template argument is F8 (fp8 operands): conv_tile_kernel<BM, BN, BK, S, LIN, SMODE, OCC, MULTI, F32, KWS, FIX, F8>
        args = name[name.index("<") + 1:name.index(">")].replace(" ", "").split(",") if "<" in name and ">" in name else []
        if len(args) >= 12 and args[11] in ("true", "1", "2"):
            return "conv fprop/dgrad fp8 (conv_tile_kernel<..., F8>)"
        return "conv fprop/dgrad (conv_tile_kernel)"
    if "wgrad_kernel" in name or "wgrad_group_kernel" in name:
        return "conv wgrad (wgrad_kernel, wgrad_group_kernel)"
    # (kernel NAMES, not argument types: roi_bwd_rows_kernel and maxpool_bwd_bnreduce_kernel take a frcnn_bn_reduce argument, which
    # appears in their mangled names)
    if "roi_fwd" in name:
        return "RoI crop+pool forward (roi_fwd_kernel)"
    if "roi_bwd" in name:
        return "RoI crop+pool backward (roi_bwd_rows_kernel)"
    if "nms_" in name:
        return "combined NMS (nms_class_kernel + nms_merge_kernel)"
    if "maxpool_bwd_bnreduce" in name:
        return "other"
    if "bn_train" in name or "bn_bwd" in name or "bn_apply" in name:
        return "batchnorm apply / reduce (bn_*_kernel)"
    return "other"


def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "")[:150]


def load_counter(path, counter):
    per = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        per[k][0] += 1
        per[k][1] += float(r["Counter_Value"])
    return per


def main():
    tag, steps, stats_csv, fetch_csv, write_csv, command = sys.argv[1], float(sys.argv[2]), sys.argv[3], sys.argv[4], sys.argv[5], sys.argv[6]
    fetch, write = load_counter(fetch_csv, "FETCH_SIZE"), load_counter(write_csv, "WRITE_SIZE")
    dur = {}
    for r in csv.DictReader(open(stats_csv)):
        dur[short(r["Name"])] = (int(r["Calls"]), float(r["TotalDurationNs"]))
    out_csv = os.path.join(ROOT, "profiles", tag + "_hbm_per_kernel.csv")
    fams = collections.defaultdict(lambda: {"launches": 0, "rd": 0.0, "wr": 0.0, "ns": 0.0, "trace_launches": 0})
    with open(out_csv, "w") as fh:
        w = csv.writer(fh)
        w.writerow(["kernel", "launches_pmc", "FETCH_SIZE_KiB_raw", "read_bytes_corrected_x2", "WRITE_SIZE_KiB", "write_bytes", "hbm_bytes_per_launch",
                    "launches_trace", "avg_duration_us", "hbm_TBps"])
        for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch[k][1] + write[k][1])):
            n = max(fetch[k][0], write[k][0])
            rd, wr = 2.0 * fetch[k][1] * 1024.0, write[k][1] * 1024.0
            calls, ns = dur.get(k, (0, 0.0))
            avg_us = ns / calls / 1e3 if calls else 0.0
            per_launch = (rd + wr) / max(n, 1)
            w.writerow([k, n, "%.0f" % fetch[k][1], "%.0f" % rd, "%.0f" % write[k][1], "%.0f" % wr, "%.0f" % per_launch, calls, "%.2f" % avg_us,
                        "%.2f" % (per_launch / (avg_us * 1e-6) / 1e12 if avg_us else 0.0)])
            f = fams[family(k)]
            f["launches"] += n
            f["rd"] += rd
            f["wr"] += wr
        for k, (calls, ns) in dur.items():
            f = fams[family(k)]
            f["ns"] += ns
            f["trace_launches"] += calls
    from bench import kernel_source_hash, workload_key_of_command
    off = {"kernel_source_hash": kernel_source_hash(), "workload": workload_key_of_command(command), "from": command, "steps": steps, "per_kernel_table": os.path.relpath(out_csv, ROOT), "families": {}}
    for name, f in fams.items():
        off["families"][name] = {
            "launches_per_step": round(f["launches"] / steps, 2) if f["launches"] else round(f["trace_launches"] / steps, 2),
            "avg_launch_us": round(f["ns"] / f["trace_launches"] / 1e3, 3) if f["trace_launches"] else None,
            "ms_per_step": round(f["ns"] / 1e6 / steps, 4),
            "hbm_read_bytes_per_step": round(f["rd"] / steps), "hbm_write_bytes_per_step": round(f["wr"] / steps),
            "hbm_bytes_per_launch": round((f["rd"] + f["wr"]) / f["launches"]) if f["launches"] else None}
    name = "%s_offline%s.json" % (tag[:3], os.environ.get("OFFLINE_VARIANT", ""))
    json.dump(off, open(os.path.join(ROOT, "profiles", name), "w"), indent=1)
    print(json.dumps(off["families"], indent=1))


if __name__ == "__main__":
    main()
