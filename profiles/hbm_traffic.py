"""HBM traffic of the conv kernel families from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), corrected as
/opt/skills/guides/MI355X_MICROARCH.md 'HBM' prescribes for gfx950: FETCH_SIZE counts 128-byte requests of wide coalesced
reads at 64 B -> doubled; WRITE_SIZE is exact for 16-B/lane stores and float atomics.  Both in KiB per dispatch.

usage: python profiles/hbm_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <steps> > out.json
"""
import collections
import csv
import json
import sys


def family(name):
    if "conv_tile_kernel" in name or "igemm_kernel" in name:
        return "conv fprop/dgrad (conv_tile_kernel, igemm_kernel)"
    if "wgrad_kernel" in name or "wgrad_group_kernel" in name:
        return "conv wgrad (wgrad_kernel, wgrad_group_kernel)"
    if "bn_" in name:
        return "batchnorm (bn_*_kernel)"
    return "other"


def load(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        f = family(r["Kernel_Name"])
        agg[f][0] += 1
        agg[f][1] += float(r["Counter_Value"])
    return agg


def main():
    fetch, write, steps = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE"), float(sys.argv[3])
    out = {}
    for f in sorted(set(fetch) | set(write)):
        n = max(fetch[f][0], write[f][0])
        rd = 2.0 * fetch[f][1] * 1024.0          # gfx950 correction: x2
        wr = write[f][1] * 1024.0
        out[f] = {"launches_per_step": n / steps, "read_bytes_per_step": rd / steps, "write_bytes_per_step": wr / steps,
                  "bytes_per_launch": (rd + wr) / max(n, 1)}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
