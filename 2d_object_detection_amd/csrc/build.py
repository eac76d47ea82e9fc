"""Build lib2dod_hip.so (the C-ABI library of include/frcnn_hip.h) for gfx950 with hipcc.

In-tree build: the .so lands next to the package so that it travels with a repo snapshot.
hipcc cross-compiles without a GPU.  Usage: python 2d_object_detection_amd/csrc/build.py [--force]
"""
import hashlib
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
LIB = os.path.join(PKG, "lib2dod_hip.so")
SOURCES = ["elementwise.hip", "conv_tile.hip", "conv_wgrad.hip", "boxes_nms.hip", "roi.hip", "targets_losses.hip", "host_io.hip", "fp8.hip", "fpn.hip"]
HEADERS = ["common.h", "conv_common.h", os.path.join("..", "..", "include", "frcnn_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function", "-DFRCNN_BUILD"]
# kernel-development variants live in their own library file and object directory (FRCNN_LIB selects it at load time):
#   FRCNN_SWEEP=1  extra tile instantiations + FRCNN_TILE / FRCNN_KWS / FRCNN_WGRAD overrides  -> lib2dod_hip_sweep.so
#   FRCNN_STAMPS=1 per-workgroup phase stamps in the conv kernel (tools/conv_stamps.py)          -> lib2dod_hip_stamps.so
VARIANT = ""
if os.environ.get("FRCNN_SWEEP"):
    FLAGS.append("-DFRCNN_SWEEP")
    VARIANT += "_sweep"
if os.environ.get("FRCNN_STAMPS"):
    FLAGS.append("-DFRCNN_STAMPS")
    VARIANT += "_stamps"
# one-off A/B builds: FRCNN_DEFINES="NAME[,NAME...]" adds -DNAME and FRCNN_TAG=xyz names the result lib2dod_hip_xyz.so (tools/ab_lib.sh)
for name in filter(None, os.environ.get("FRCNN_DEFINES", "").split(",")):
    FLAGS.append("-D" + name)
if os.environ.get("FRCNN_TAG"):
    VARIANT += "_" + os.environ["FRCNN_TAG"]
if VARIANT:
    LIB = os.path.join(PKG, "lib2dod_hip%s.so" % VARIANT)


def source_hash():
    """sha1 over the HIP sources, their headers and the C-ABI header (12 hex digits): what frcnn_source_hash() returns and what
    bench.py keys its committed rocprof profiles with (bench.kernel_source_hash is this function)."""
    h = hashlib.sha1()
    for name in sorted(os.listdir(HERE)):
        if name.endswith((".hip", ".h")):
            h.update(open(os.path.join(HERE, name), "rb").read())
    h.update(open(os.path.join(HERE, "..", "..", "include", "frcnn_hip.h"), "rb").read())
    return h.hexdigest()[:12]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "_obj" + VARIANT)
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(HERE, h) for h in HEADERS] + [os.path.abspath(__file__)]
    jobs = []
    objs = []
    digest = source_hash()
    stamp = os.path.join(objdir, "source_hash.txt")
    stale_hash = not os.path.exists(stamp) or open(stamp).read().strip() != digest
    for s in SOURCES:
        src = os.path.join(HERE, s)
        obj = os.path.join(objdir, s.replace(".hip", ".o"))
        objs.append(obj)
        extra = []
        if s == "host_io.hip":                   # carries frcnn_source_hash(): rebuilt whenever any source changed
            extra = ['-DFRCNN_SOURCE_HASH="%s"' % digest]
        if force or _newer(obj, [src] + hdrs) or (extra and stale_hash):
            jobs.append([hipcc] + FLAGS + extra + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _newer(LIB, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    with open(stamp, "w") as fh:
        fh.write(digest + "\n")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
