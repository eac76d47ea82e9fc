"""Headline benchmark: images/sec of the full Faster-RCNN training step (ResNet-50, 375x1242
synthetic KITTI batches, bf16) on N MI355X GPUs of one node.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" = forward + RPN NMS + RoI pooling + target assignment + sampling + losses + backward +
SGD-momentum + prediction NMS for one batch already resident in HBM (the reference's
FasterRCNN.train_step, models/faster_rcnn.py:59-117).  Prints ONE JSON line on rank 0:

  value / ms_per_step   W untimed warm-up steps, then EXACTLY K steps between barrier + synchronize, MAX over ranks
  windows               the same K-step region timed `--windows` times back to back (the first one is `value`): min / median / max
  roofline              dominant kernel family (conv fprop/dgrad, bf16 MFMA): algorithmic FLOP / live HIP-event launch durations;
                        `families`: every kernel family of the step with its time and its TFLOP/s or algorithmic GB/s;
                        `rocprof`: the same quotient from the rocprofv3 summary committed under profiles/ (+ PMC HBM traffic)
  cpu_baseline          the CPU oracle (kind "port") timed on this host: SURVEY 8(d) protocol, see cpu_baseline()
  other_configs         (default N = 1 run only) BASELINE.json configs[3], configs[4] and the reference's own configuration (config.json:
                        600x1987, batch 2: "reference_default") timed after the headline's region -- one discarded window, then 3 whose
                        median is their `value` -- with their own roofline; the headline's value / config / roofline do not depend on them
  other_configs_summary the LAST key of the line: {name: [images/s, ms/step, roofline frac]} (survives a truncated tail)

Every step of a window trains on the next of `--resident-batches` (8) synthetic batches resident in HBM (seeds 1234 + rank + 100 i):
the head cannot memorise one batch, so the density of the RoI-backward gradient stays that of a real run.
`python bench.py --gpus N` with N > 1 and no torchrun environment launches its own N ranks (a child `python -m torch.distributed.run`,
started before this process touches the GPU) and relays rank 0's line.
"""
import argparse
import copy
import hashlib
import importlib
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch

METRIC = "images/sec training, ResNet-50 Faster-RCNN KITTI 1242x375, 1/2/4/8 GPU"
PEAK_BF16_TFLOPS = 2500.0          # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_FP8_TFLOPS = 5000.0           # dense fp8 (block-scaled f8f6f4 MFMA at K = 128), same guide
PEAK_HBM_GBS = 8000.0              # HBM3E spec (6.3 TB/s achievable, same guide)
FAM_CONV = "conv fprop/dgrad (conv_tile_kernel)"
FAM_CONV_F8 = "conv fprop/dgrad fp8 (conv_tile_kernel<..., F8>)"
FAM_WGRAD = "conv wgrad (wgrad_kernel, wgrad_group_kernel)"


def kernel_source_hash():
    """sha1 over the HIP sources + headers: ties an offline profile (profiles/*.json) to the code it was taken from -- the hash the
    library itself reports (frcnn_source_hash(), csrc/build.py)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("_frcnn_build", os.path.join(ROOT, "2d_object_detection_amd", "csrc", "build.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.source_hash()


def conv_flops(d, true_cin=None, true_cout=None):
    """Algorithmic FLOPs (2*MAC) of one implicit-GEMM launch with its unpadded channel counts."""
    m = d.n * d.ho * d.wo
    k = d.kh * d.kw * (true_cin if true_cin is not None else d.cin)
    return 2.0 * m * k * (true_cout if true_cout is not None else d.cout)


RPN_HEAD_ROWS = 72                 # real rows of the merged RPN head bank: 6 x 12 anchors per location (feature pyramid: 6 x 3 = 18)


def _true_dims(d):
    """un-padded channel counts: stem taps 7x(8x4) carry 7x7x3 real values; RPN heads 72 (18) of 128; RCNN heads 36 of 64"""
    cin, cout = d.cin, d.cout
    if d.in_pix_stride == 4 and d.cin == 32:
        return 21.0, cout
    if d.kh == 1 and d.cin == 256 and d.cout == 128:
        cout = RPN_HEAD_ROWS
    if d.kh == 1 and d.cin == 128 and d.cout == 256:
        cin = RPN_HEAD_ROWS
    if d.cout == 64 and d.cin > 4096:
        cout = 36
    if d.cin == 64 and d.cout > 4096:
        cin = 36
    return cin, cout


def _tensor_bytes(args, kwargs, skip=()):
    """Sum of the sizes of the distinct large tensors a launch touches: the algorithmic bytes of an elementwise / gather
    kernel that reads or writes each of its operands once."""
    seen, total = set(), 0
    for i, a in enumerate(list(args) + list(kwargs.values())):
        if torch.is_tensor(a) and a.numel() * a.element_size() >= 65536 and a.data_ptr() not in seen and i not in skip:
            seen.add(a.data_ptr())
            total += a.numel() * a.element_size()
    return float(total)


def kernels_of(ops, fn, args, kwargs):
    """GPU kernels one plan call launches (what a rocprofv3 trace counts): a grouped weight-gradient call is one kernel per addressing
    mode of its group, a combined NMS over more than one class is the per-class kernel + the merge kernel."""
    if fn is ops.conv2d_wgrad_grouped:
        return max(1, len([p for p in ops.conv2d_wgrad_describe(group=args[0]).split("; ") if p.strip()]))
    if getattr(fn, "__name__", "") in ("nms_combined", "nms_combined_abs"):
        return 1 if int(args[5]) == 1 else 2
    return 1


def classify(ops, fn, args, kwargs):
    """(family name, algorithmic FLOP, algorithmic bytes) of one plan launch."""
    if fn is ops.conv2d_fprop or fn is ops.conv2d_dgrad_bnreduce or fn is ops.conv2d_fprop_bnin:      # (bnin: + the input layer's BatchNorm, not counted)
        return FAM_CONV, conv_flops(args[0], *_true_dims(args[0])), 0.0
    if fn is ops.conv2d_fprop_fp8 or fn is ops.conv2d_dgrad_fp8:
        return FAM_CONV_F8, conv_flops(args[0], *_true_dims(args[0])), 0.0
    if fn is ops.conv2d_wgrad or fn is ops.conv2d_wgrad_fp8:
        return FAM_WGRAD, conv_flops(args[0], *_true_dims(args[0])), 0.0
    if fn is ops.conv2d_wgrad_grouped:
        return FAM_WGRAD, sum(conv_flops(it[0], *_true_dims(it[0])) for it in args[0].items), 0.0
    name = getattr(fn, "__name__", str(fn))
    if name in ("bn_train_apply", "bn_apply", "bn_bwd_apply_fused", "bn_bwd_apply_fused_red2", "bn_bwd_reduce", "bn_train_apply_maxpool", "bn_train_apply_dual"):
        return "batchnorm apply / reduce (bn_*_kernel)", 0.0, _tensor_bytes(args, kwargs)
    if name in ("roi_crop_pool_fwd", "roi_crop_pool_fwd_level"):
        return "RoI crop+pool forward (roi_fwd_kernel)", 0.0, _tensor_bytes(args, kwargs)
    if name in ("roi_crop_pool_bwd_bf16", "roi_crop_pool_bwd_bf16_level", "roi_crop_pool_bwd_bf16_add"):
        return "RoI crop+pool backward (roi_bwd_rows_kernel)", 0.0, _tensor_bytes(args, kwargs)
    if name in ("nms_combined", "nms_combined_abs"):
        # boxes [B,N,q,4] + scores [B,N,*] in, padded outputs out (the workspace is scratch; _abs: + the kept boxes once more, 16 B each)
        return "combined NMS (nms_class_kernel + nms_merge_kernel)", 0.0, float(sum(t.numel() * t.element_size() for t in (args[0], args[1], args[12], args[13], args[14])))
    if name in ("assign_targets", "sample_indices", "losses", "losses_head_grad", "losses_rpn_head_grad", "rpn_head_grad", "rcnn_head_grad",
                "rpn_head_post", "rpn_head_post_decode", "rcnn_head_post", "decode_boxes", "boxes_scale", "rpn_head_post_level", "rpn_head_grad_level",
                "roi_assign_levels"):
        return "targets / sampling / losses / head post", 0.0, _tensor_bytes(args, kwargs)
    if name in ("sgd_momentum", "sgd_momentum_fused"):
        return "SGD-momentum update (sgd_fused_kernel)", 0.0, float(args[4]) * (4 + 4 + 4 + 4 + 4 + 2)
    return "other (pool, preprocess, re-layouts, fills, column sums)", 0.0, _tensor_bytes(args, kwargs)


def profile_kernels(model, built, steps=3):
    """Eager replay of the train plan with HIP events (torch events on the launch stream) around EVERY launch.
    Returns per-family {launches, seconds, flops, bytes} per step."""
    ops = importlib.import_module("2d_object_detection_amd.ops")
    plan = built["plan"]
    records = []
    for seg in plan.segments:
        for fn, args, kwargs, _branch in seg:
            if fn is None:                       # join marker of a side-stream branch (the profile pass runs serially)
                continue
            name, fl, by = classify(ops, fn, args, kwargs)
            records.append((fn, args, kwargs, name, fl, by, kernels_of(ops, fn, args, kwargs)))
    state = model._snapshot(built["optimizer"])
    # cost of an event pair itself (two marker packets back to back, nothing between): subtracted from every measurement
    torch.cuda._sleep(20_000_000)
    cal = []
    for _ in range(200):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        e1.record()
        cal.append((e0, e1))
    torch.cuda.synchronize()
    pair_overhead_s = sorted(a.elapsed_time(b) for a, b in cal)[len(cal) // 2] * 1e-3
    events = []
    for it in range(steps + 1):
        # Let the host run ahead: a ~30 ms device-side spin is enqueued first, so every event record and launch of the step
        # is already queued when the GPU reaches it.  Otherwise each event pair would also time the 5-10 us the stream
        # idles between an eager launch and the start of its kernel (a graph replay has no such gaps).
        torch.cuda._sleep(60_000_000)
        plan._zero_prologue()                    # (the plan's one zero-fill launch, not timed)
        for fn, args, kwargs, name, fl, by, nk in records:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            fn(*args, **kwargs)
            e1.record()
            if it > 0:
                events.append((name, fl, by, e0, e1, nk))
    torch.cuda.synchronize()
    # RoI backward with a DENSE pooled gradient: the kernel skips zero gradient words, and how many there are depends on what the
    # head has learnt (a saturated softmax zeroes whole rows) -- the dense time is its data-independent upper bound
    dense = {}
    roi_b = [(fn, args, kwargs) for fn, args, kwargs, name, fl, by, nk in records if name.startswith("RoI crop+pool backward")]
    if roi_b:
        gp = roi_b[0][1][0]
        keep = gp.clone()
        nz_frac = float((keep.view(torch.int16) != 0).float().mean())
        gp.copy_((torch.randn(gp.shape, device=gp.device) * 1e-3).to(gp.dtype))
        torch.cuda._sleep(20_000_000)
        pairs = []
        for _ in range(10):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for fn, args, kwargs in roi_b:
                fn(*args, **kwargs)
            e1.record()
            pairs.append((e0, e1))
        torch.cuda.synchronize()
        gp.copy_(keep)
        per = sorted(a.elapsed_time(b) * 1e3 for a, b in pairs)
        dense = {"roi_bwd_dense_us": round(max(per[len(per) // 2] - pair_overhead_s * 1e6, 0.0), 1), "roi_bwd_launches": len(roi_b),
                 "roi_bwd_nonzero_fraction_in_step": round(nz_frac, 4)}
    model._restore(state, built["optimizer"])
    if os.environ.get("FRCNN_LAYER_TABLE"):
        write_layer_table(os.environ["FRCNN_LAYER_TABLE"], ops, records, events, steps, pair_overhead_s)
    fam = {}
    for name, fl, by, e0, e1, nk in events:
        f = fam.setdefault(name, {"launches": 0, "plan_calls": 0, "seconds": 0.0, "flops": 0.0, "bytes": 0.0})
        f["launches"] += nk                      # GPU kernels (a trace's count); plan_calls: C-ABI calls = event pairs
        f["plan_calls"] += 1
        f["seconds"] += max(e0.elapsed_time(e1) * 1e-3 - pair_overhead_s, 0.0)
        f["flops"] += fl
        f["bytes"] += by
    for f in fam.values():
        for k in f:
            f[k] /= steps
        f["event_pair_overhead_us"] = pair_overhead_s * 1e6
    for k, f in fam.items():
        if k.startswith("RoI crop+pool backward"):
            f.update(dense)
    return fam


def launch_bytes(ops, fn, args, kwargs):
    """(operand bytes of the bare convolution, ALL bytes the launch must move) of one conv / weight-gradient plan call: the second adds what
    the fused epilogue or prologue reads and writes -- the residual (+ its mask), the consumer BatchNorm's z and mask of a fused
    backward reduce, the activation and mask an absorbed forward BatchNorm writes -- which round 4's table left out."""
    if fn is ops.conv2d_wgrad_grouped:
        base = full = 0.0
        for it in args[0].items:
            d = it[0]
            m, es = d.n * d.ho * d.wo, (1.0 if len(it) > 4 else 2.0)
            b_ = es * (d.n * d.hi * d.wi * d.cin + m * d.cout) + 4.0 * d.cout * d.kh * d.kw * d.cin
            base += b_
            full += b_
        return base, full
    d = args[0]
    m = d.n * d.ho * d.wo
    m_in = d.n * d.hi * d.wi
    if fn in (ops.conv2d_wgrad, ops.conv2d_wgrad_fp8):
        es = 1.0 if fn is ops.conv2d_wgrad_fp8 else 2.0
        b_ = es * (m_in * min(d.cin, d.in_pix_stride) + m * d.cout) + 4.0 * d.cout * d.kh * d.kw * d.cin
        return b_, b_
    es_in = 1.0 if fn in (ops.conv2d_fprop_fp8, ops.conv2d_dgrad_fp8) else 2.0
    m_out = d.n * d.out_h * d.out_w if getattr(d, "out_scatter", 1) > 1 else m
    out_es = 4.0 if (d.flags & (ops.CONV_OUT_F32 | ops.CONV_SPLITK_ATOMIC)) else 2.0
    base = es_in * (min(m_in, m * d.kh * d.kw) * min(d.cin, d.in_pix_stride) + d.cout * d.kh * d.kw * d.cin) + out_es * m * d.cout
    full = base
    if kwargs.get("res") is not None:
        full += 2.0 * m * d.cout + (m * d.cout / 8.0 if kwargs.get("res_mask") is not None else 0.0)
    red = kwargs.get("red") if fn is not ops.conv2d_dgrad_bnreduce else args[4]
    if red is not None:
        full += 2.0 * m_out * d.cout + m_out * d.cout / 8.0       # the consumer layer's z rows and ReLU mask bytes
    if fn is ops.conv2d_fprop_bnin:
        full += 2.0 * m_in * d.cin + m_in * d.cin / 8.0           # the absorbed BatchNorm's activation and mask stores
    return base, full


def write_layer_table(path, ops, records, events, steps, pair_overhead_s):
    """Per-launch table (median over the profiled steps) of the MFMA families: shape, us, TFLOP/s, roofline time max(FLOP / MFMA peak,
    bytes / HBM) with the launch's FULL byte count (launch_bytes) priced at the 8.0 TB/s spec and at the 6.3 TB/s the guide calls
    achievable, and the kernel that ran.  Last lines: sum of rooflines / sum of measured times."""
    n = len(records)
    tot = {"us": 0.0, "roof8": 0.0, "roof63": 0.0, "roof_min": 0.0}
    with open(path, "w") as fh:
        for j, (fn, args, kwargs, name, fl, by, nk) in enumerate(records):
            if name not in (FAM_CONV, FAM_CONV_F8, FAM_WGRAD):
                continue
            per = sorted(events[it * n + j][3].elapsed_time(events[it * n + j][4]) * 1e3 for it in range(steps))
            us = max(per[len(per) // 2] - pair_overhead_s * 1e6, 0.1)
            d = args[0] if fn is not ops.conv2d_wgrad_grouped else args[0].items[0][0]
            m = d.n * d.ho * d.wo
            base, full = launch_bytes(ops, fn, args, kwargs)
            pk = 5.0e15 if name == FAM_CONV_F8 else 2.5e15
            roof8, roof63, roof_min = max(fl / pk, full / 8e12) * 1e6, max(fl / pk, full / 6.3e12) * 1e6, max(fl / pk, base / 8e12) * 1e6
            for k_, v_ in (("us", us), ("roof8", roof8), ("roof63", roof63), ("roof_min", roof_min)):
                tot[k_] += v_
            inst = ""
            if fn is ops.conv2d_fprop or fn is ops.conv2d_dgrad_bnreduce or fn is ops.conv2d_fprop_bnin:
                inst = ops.conv2d_describe(d, fn is ops.conv2d_dgrad_bnreduce) + (" +bnin" if fn is ops.conv2d_fprop_bnin else "")
            if fn is ops.conv2d_fprop_fp8:
                inst = ops.conv2d_describe_fp8(d)
            if fn is ops.conv2d_dgrad_fp8:
                inst = ops.conv2d_describe_dgrad_fp8(d, kwargs.get("red") is not None)
            if fn is ops.conv2d_wgrad_grouped:
                inst = "%d layers" % len(args[0].items)
            fh.write("%-12s M=%7d cin=%5d cout=%5d k=%dx%d s=%d  %8.1f us %7.1f TF/s  bytes %7.1f MB (conv operands %7.1f)  roof %6.1f us @8.0 / %6.1f @6.3 TB/s (%s)  "
                     "frac %.2f / %.2f  %s\n" % (
                         "wgrad" if name == FAM_WGRAD else "fp8" if name == FAM_CONV_F8 else "fprop/dgrad", m, d.cin, d.cout, d.kh, d.kw, d.stride, us,
                         fl / us / 1e6, full / 1e6, base / 1e6, roof8, roof63, "mfma" if fl / pk > full / 6.3e12 else "hbm", roof8 / us, roof63 / us, inst))
        fh.write("SUM measured %.1f us; sum of rooflines: %.1f us at 8.0 TB/s (ratio %.3f), %.1f us at 6.3 TB/s (ratio %.3f); with the bare conv operands "
                 "only (round 4's column) %.1f us (ratio %.3f)\n" % (tot["us"], tot["roof8"], tot["roof8"] / tot["us"], tot["roof63"], tot["roof63"] / tot["us"],
                                                                      tot["roof_min"], tot["roof_min"] / tot["us"]))


def usable_cores():
    """CPU cores this process may really use (affinity mask, cgroup quota): torch defaults to the host's core count."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(cfg, budget_s=150.0):
    """The CPU oracle (fp32 restatement of the reference path: PyTorch-CPU + C NMS, kind "port") timed on this host at
    BASELINE config 1 (batch 1, 375x1242), SURVEY 8(d) protocol: forward-only and full-train-step legs, each with
    torch.set_num_threads(all usable cores) and (1); all-core legs 3 warm-up + 10 timed steps, median.  The single-thread
    legs are a bounded sample (1 warm-up + 3 timed) so that the default run stays within minutes; `value` is the all-core
    train-step leg.  A leg that would overrun the remaining budget is cut short and says so."""
    from oracle import faster_rcnn as O
    images, gl, gb = O.synthetic_batch(1, cfg["image_shape"], seed=1234)
    cores = usable_cores()
    t_start = time.perf_counter()
    legs = {}

    def run(leg, threads, warm, timed):
        torch.set_num_threads(threads)
        torch.manual_seed(0)
        p = O.init_params(cfg, seed=0)
        vel = {}
        times = []
        for s in range(warm + timed):
            t0 = time.perf_counter()
            if leg == "forward":
                with torch.no_grad():
                    O.forward(p, cfg, images, True)
            else:
                O.train_step(p, vel, cfg, images, gl, gb, lr=1e-5, step=s, seed=0)
            dt = time.perf_counter() - t0
            if s >= warm:
                times.append(dt)
            if time.perf_counter() - t_start > budget_s and len(times) >= 1:
                break
        times.sort()
        med = times[len(times) // 2]
        return {"threads": threads, "warmup": warm, "timed": len(times), "median_s": round(med, 4), "min_s": round(times[0], 4),
                "max_s": round(times[-1], 4), "images_per_s": round(1.0 / med, 4), "cut_short": len(times) < timed}

    legs["forward_all_cores"] = run("forward", cores, 3, 10)
    legs["train_step_all_cores"] = run("train", cores, 3, 10)
    legs["forward_1_thread"] = run("forward", 1, 1, 3)
    legs["train_step_1_thread"] = run("train", 1, 1, 3)
    torch.set_num_threads(cores)
    head = legs["train_step_all_cores"]
    return {"value": head["images_per_s"], "unit": "images/s", "cores": cores, "kind": "port",
            "sample": "oracle (PyTorch-CPU fp32 + C NMS) full train step, batch 1, 375x1242, %d threads, %d warm-up + %d timed steps, median %.2f s/step; "
                      "forward-only %.2f s; 1 thread: train step %.2f s, forward %.2f s (bounded sample: 1 warm-up + %d timed)" % (
                          cores, head["warmup"], head["timed"], head["median_s"], legs["forward_all_cores"]["median_s"],
                          legs["train_step_1_thread"]["median_s"], legs["forward_1_thread"]["median_s"], legs["train_step_1_thread"]["timed"]),
            "legs": legs}


DEFAULT_SHAPE = (375, 1242)        # BASELINE.json's image size (the reference's own config.json:3 says 600 x 1987)


def workload_key(depth, batch, proposals, fp8, fpn, shape=None):
    """What a trace's per-launch figures belong to: the step's shapes (network, batch, proposals, precision, topology, image size)."""
    key = "R%d,b%d,P%d,%s,%s" % (int(depth), int(batch), int(proposals) or 300, "fp8" if fp8 else "bf16", "fpn" if fpn else "c4")
    if shape and tuple(shape[:2]) != DEFAULT_SHAPE:
        key += ",%dx%d" % (shape[0], shape[1])
    return key


def workload_key_of_command(cmd):
    """workload_key of a recorded `... bench.py <flags>` command line (profiles/hbm_traffic.py stores it with the trace)."""
    tok = cmd.split()

    def val(flag, default):
        return int(tok[tok.index(flag) + 1]) if flag in tok else default
    shape = (int(tok[tok.index("--image-shape") + 1]), int(tok[tok.index("--image-shape") + 2])) if "--image-shape" in tok else None
    return workload_key(val("--depth", 50), val("--batch-per-gpu", 4), val("--proposals", 0), "--fp8" in tok, "--fpn" in tok, shape)


def offline_profile(family, variant="", workload=None):
    """rocprofv3 numbers for `family` committed under profiles/ (kernel-trace summary + PMC HBM traffic of this same command),
    valid only for the kernel sources AND the workload they were taken from: the newest profiles/r*_offline<variant>.json whose
    source hash is this tree's and whose recorded command ran the same shapes (a batch-8 trace says nothing about a batch-4
    launch).  Returns (family record or None, provenance / reason string)."""
    import glob
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_offline%s.json" % variant)), reverse=True)
    if not cands:
        return None, "no profiles/r*_offline%s.json" % variant
    mine = kernel_source_hash()
    stale = []
    for path in cands:
        off = json.load(open(path))
        theirs = off.get("workload") or workload_key_of_command(off.get("from", ""))
        if workload is not None and theirs != workload:
            stale.append("%s is a trace of %s, not of %s" % (os.path.basename(path), theirs, workload))
            continue
        if off.get("kernel_source_hash") == mine:
            return off.get("families", {}).get(family), "offline: profiles/%s (%s) @ kernel sources %s" % (
                os.path.basename(path), off.get("from", "?"), mine)
        stale.append("%s@%s" % (os.path.basename(path), off.get("kernel_source_hash")))
    return None, "no committed trace for this run: %s (this tree's kernel sources: %s)" % ("; ".join(stale), mine)


BASE_SPEC = {"depth": 50, "batch": 4, "proposals": 0, "fp8": False, "fpn": False, "shape": None}
OTHER_CONFIGS = {
    "configs[3]": {"depth": 101, "batch": 2, "proposals": 1000, "fp8": False, "fpn": False, "shape": None},
    "configs[4]": {"depth": 50, "batch": 8, "proposals": 0, "fp8": True, "fpn": True, "shape": None},
    # the reference's OWN configuration: /root/reference config.json:3 image_shape [600, 1987, 3], train_faster_rcnn.py:52-54 batch 2
    "reference_default": {"depth": 50, "batch": 2, "proposals": 0, "fp8": False, "fpn": False, "shape": (600, 1987)},
}


def workload_text(spec, world):
    depth, B, P, fp8, fpn = spec["depth"], spec["batch"], spec["proposals"] or 300, spec["fp8"], spec["fpn"]
    if spec.get("shape") and tuple(spec["shape"][:2]) != DEFAULT_SHAPE:
        return ("ResNet-%d(C4) Faster-RCNN full train step, %s, batch %d per GPU, %dx%d synthetic batches (the reference's own config.json "
                "image_shape, not BASELINE.json's 375x1242), %d proposals, 7 classes" % (depth, "fp8" if fp8 else "bf16", B, spec["shape"][0], spec["shape"][1], P))
    if fp8:
        return ("ResNet-%d%s Faster-RCNN full train step, fp8: e4m3 weights (per output channel) and activations (per tensor, delayed "
                "scaling), e5m2 gradients, in the forward convolutions, data gradients and weight gradients of the backbone wherever the channel counts allow (K >= 256 for forward / data gradients)%s on "
                "the f8f6f4 MFMA path; bf16 storage and remaining layers; batch %d per GPU, 375x1242 synthetic KITTI, "
                "%d proposals, 7 classes (BASELINE.json configs[4]%s)" % (
                    depth, "-FPN (pyramid over C2..C4, RPN on P2..P5, per-level RoI heads)" if fpn else "(C4)",
                    "" if fpn else " and of the RPN's 3x3", B, P,
                    ("" if B == 8 else ": its topology and precision at another batch") if fpn else
                    "'s precision%s; C4 backbone, no FPN" % (" and batch" if B == 8 else "")))
    if fpn:
        return ("ResNet-%d-FPN Faster-RCNN full train step, bf16, batch %d per GPU, 375x1242 synthetic KITTI, %d proposals, 7 classes "
                "(BASELINE.json configs[4]'s topology in bf16)" % (depth, B, P))
    return ("ResNet-%d(C4) Faster-RCNN full train step, bf16, batch %d per GPU, 375x1242 synthetic KITTI, "
            "%d proposals, 7 classes (BASELINE.json configs[%d])" % (depth, B, P, 3 if depth == 101 else 1 if world == 1 else 2))


def measure(spec, args, rank, world, dev, windows, with_segmented, with_rccl_leg, discard=0):
    """Build the model of `spec`, warm up, time `windows` regions of exactly args.steps steps (barrier + synchronize on both sides, MAX
    over ranks) and -- on rank 0 -- attach the per-family kernel profile.  Returns the result dict (the headline's shape).
    discard (other_configs only): that many regions are timed and reported but not counted first, and `value` is the MEDIAN of the
    `windows` that follow -- the first region after another model's teardown measured 5.6 % slow on configs[4];
    the headline (discard 0) keeps the contract's rule: `value` is its FIRST region."""
    global RPN_HEAD_ROWS
    RPN_HEAD_ROWS = 18 if spec["fpn"] else 72
    D = importlib.import_module("2d_object_detection_amd.distributed")
    M = importlib.import_module("2d_object_detection_amd.models.faster_rcnn")
    OPT = importlib.import_module("2d_object_detection_amd.optimizers")
    C = importlib.import_module("2d_object_detection_amd.config")
    DATA = importlib.import_module("2d_object_detection_amd.data")
    import torch.distributed as dist

    # 375 x 1242 (BASELINE.json) unless the spec names another image size; 7 classes, reference hyper-parameters
    cfg = C.default_config((spec["shape"][0], spec["shape"][1], 3)) if spec.get("shape") else C.default_config()
    if spec["proposals"]:
        cfg["rpn"]["nms"]["max_total_size"] = cfg["rpn"]["nms"]["max_output_size_per_class"] = spec["proposals"]
    B = spec["batch"]
    model = M.FasterRCNN(cfg, depth=spec["depth"], device=dev, seed=0, sampling_seed=rank, world_size=world,     # (per-rank fg/bg sample positions)
                         precision="fp8" if spec["fp8"] else "bf16", topology="fpn" if spec["fpn"] else "c4")
    model.use_graphs = not args.no_graphs
    # Reference schedule shape (train_faster_rcnn.py:62-68: boundaries 40k/80k), scaled by --lr-scale: the reference's
    # 1e-3 presumes ImageNet-pretrained weights; with the seeded random init used here (no network) and the un-normalised
    # regression loss it diverges within ~10 steps, which would make the timed workload degenerate (NaN boxes).
    sc = args.lr_scale
    opt = OPT.SGD(learning_rate=OPT.PiecewiseConstantDecay([40000, 80000], [1e-3 * sc, 1e-4 * sc, 1e-5 * sc]), momentum=0.9)

    # synthetic KITTI-like batches, resident in HBM, per-rank seeds; step s of a region trains on batch s mod NB
    NB = max(1, args.resident_batches)
    batches = [DATA.synthetic_batch(B, cfg["image_shape"], seed=1234 + rank + 100 * i, device=dev) for i in range(NB)]

    sync = D.GradientSynchronizer(model.store.g, model.store.buckets) if world > 1 else None
    hook = [sync.after_segment if sync is not None else None]

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_window():
        barrier()
        t0 = time.perf_counter()
        for s in range(args.steps):
            out = model.train_step(*batches[s % NB], opt, sync_fn=hook[0])     # (`hook` is read at call time: the extra legs swap it)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
            dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt, out

    for s in range(args.warmup):
        losses, preds = model.train_step(*batches[s % NB], opt, sync_fn=hook[0])
    # every timed window starts from the SAME weights, momentum, moving statistics, fp8 scales and step counter (the state after the
    # warm-up): window 5 times what window 1 timed, and the synthetic run cannot drift towards non-finite boxes while it is measured
    barrier()
    state0 = model._snapshot(opt)

    def window():
        model._restore(state0, opt)
        return timed_window()

    discarded_ms = [window()[0] / args.steps * 1e3 for _ in range(discard)]
    dt, (losses, preds) = window()                             # the contract's region: EXACTLY K steps, barrier + synchronize both sides
    loss_vals = {k: float(v) for k, v in losses.items()}
    window_ms = [dt / args.steps * 1e3]
    for _ in range(max(0, windows - 1)):                       # more evidence than one 0.1 s region: the same region again
        window_ms.append(window()[0] / args.steps * 1e3)
    srt = sorted(window_ms)
    if discard:
        dt = srt[len(srt) // 2] * 1e-3 * args.steps            # (other_configs: the median region)
    ms = dt / args.steps * 1e3
    value = world * B * args.steps / dt
    # what data parallelism costs before any byte moves: the same step as its backward-segment graphs with a (no-op) hook between
    # them -- the form every rank runs when a GradientSynchronizer interleaves bucket all-reduces -- against the one-graph replay
    segmented_ms = None
    if world == 1 and model.use_graphs and with_segmented:
        hook_calls = []
        keep_hook, hook[0] = hook[0], (lambda i, n: hook_calls.append(i))
        try:
            segmented_ms = sorted(window()[0] / args.steps * 1e3 for _ in range(3))[1]
        finally:
            hook[0] = keep_hook
        assert len(hook_calls) >= 3 * args.steps, "the segmented run did not call the hook"
    # ... and with the collectives really issued: a world-1 process group on the backend a multi-GPU run uses (nccl = RCCL), one
    # dist.all_reduce per gradient bucket on the comm stream between the segment replays, the update segment waiting for them
    rccl = None
    if world == 1 and model.use_graphs and with_rccl_leg:
        try:
            D.init_from_env(force=True)
            fs = D.GradientSynchronizer(model.store.g, model.store.buckets, force=True)
            keep_hook, hook[0] = hook[0], fs.after_segment
            try:
                leg = sorted(window()[0] / args.steps * 1e3 for _ in range(3))[1]
            finally:
                hook[0] = keep_hook
            rccl = {"ms_per_step": round(leg, 4), "all_reduces_per_step": fs.calls / (3.0 * args.steps), "backend": dist.get_backend(),
                    "bytes_per_step": fs.bytes_per_step,
                    "note": "world 1: the call path (comm stream, ready events, segment graphs) runs; a one-rank all-reduce moves no bytes over xGMI"}
        except Exception as e:                                 # (never fail the headline on the rehearsal leg)
            rccl = {"error": "%s: %s" % (type(e).__name__, e)}
        # ... and with the data-parallel step captured as ONE graph, its all-reduces inside (FasterRCNN.capture_collectives): first in a
        # CHILD process under a time-out (tests/_dp_capture_child.py: a small model; a stream capture RCCL refuses raises there, one that
        # hangs is killed there), then -- only if the child succeeded -- the same leg on this model
        if isinstance(rccl, dict) and "error" not in rccl:
            rccl["captured"] = captured_collectives_leg(model, window, hook, D, args)

    out = {
        "metric": METRIC, "value": round(value, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "fp8" if spec["fp8"] else "bf16",
        "data": "synthetic",
        "config": {"workload": workload_text(spec, world),
                   "global_batch": world * B, "image_shape": cfg["image_shape"], "parallelism": "dp%d" % world,
                   "hip_graphs": model.use_graphs, "kernel_launches_per_step": model._train_plan["plan"].num_launches,
                   "resident_batches": NB, "grad_compress": getattr(sync, "compress", None),
                   "segmented_ms_per_step": None if segmented_ms is None else round(segmented_ms, 4),
                   "forced_collectives_world1": rccl,
                   "segments": len(model._train_plan["plan"].segments)},
        "windows": {"steps_each": args.steps, "ms_per_step": [round(x, 4) for x in window_ms],
                    "value_is": "median of these regions (after the discarded ones)" if discard else "the first of these regions",
                    "discarded_first_ms_per_step": [round(x, 4) for x in discarded_ms], "min": round(srt[0], 4),
                    "median": round(srt[len(srt) // 2], 4), "max": round(srt[-1], 4),
                    "images_per_s_median": round(world * B / (srt[len(srt) // 2] * 1e-3), 2)},
        "final_losses": loss_vals,
    }
    if spec["fp8"]:
        out["config"]["fp8_status"] = model.fp8_status()
    if rank == 0:
        fam = profile_kernels(model, model._train_plan, args.profile_steps) if args.profile_steps > 0 else {}
        if fam:
            attach_roofline(out, fam, spec, ms)
    out["_cfg"] = cfg
    del model, opt, batches
    torch.cuda.empty_cache()
    return out


def captured_collectives_leg(model, window, hook, D, args):
    child = os.path.join(ROOT, "tests", "_dp_capture_child.py")
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    try:
        proc = subprocess.Popen([sys.executable, child], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        try:
            text, _ = proc.communicate(timeout=180)
        except subprocess.TimeoutExpired:
            proc.kill()
            proc.communicate()
            return {"error": "probe child did not finish within 180 s: captured form not attempted in this process"}
        ok = [ln for ln in text.splitlines() if ln.startswith("DP_CAPTURE_OK")]
        if proc.returncode != 0 or not ok:
            return {"error": "probe child failed (rc %d): %s" % (proc.returncode, text[-300:])}
        probe = json.loads(ok[0][len("DP_CAPTURE_OK "):])
        fs = D.GradientSynchronizer(model.store.g, model.store.buckets, force=True)
        keep_hook, keep_flag = hook[0], model.capture_collectives
        hook[0], model.capture_collectives = fs.after_segment, True
        try:
            leg = sorted(window()[0] / args.steps * 1e3 for _ in range(3))[1]
            dp = model._train_plan.get("dp") or {}
        finally:
            hook[0], model.capture_collectives = keep_hook, keep_flag
        return {"ms_per_step": round(leg, 4), "one_graph": dp.get("graph") is not None, "fallback_reason": dp.get("error"),
                "all_reduces_per_step": fs.calls / (3.0 * args.steps), "probe_small_model_ms": probe,
                "note": "the step and its bucket all-reduces replayed as ONE hipGraph (FRCNN_CAPTURE_COLLECTIVES; opt-in for N > 1)"}
    except Exception as e:
        return {"error": "%s: %s" % (type(e).__name__, e)}


def attach_roofline(out, fam, spec, ms):
    # the headline kernel family: the bf16 MFMA convolutions; with fp8 the fp8 convolutions (priced against the fp8 peak)
    dom = FAM_CONV_F8 if (spec["fp8"] and FAM_CONV_F8 in fam) else FAM_CONV if FAM_CONV in fam else max(fam, key=lambda k: fam[k]["seconds"])
    f = fam[dom]
    peak = PEAK_FP8_TFLOPS if dom == FAM_CONV_F8 else PEAK_BF16_TFLOPS
    gflop_per_launch = f["flops"] / f["launches"] / 1e9
    # HEADLINE = the figure a reader can reproduce from profiles/: algorithmic FLOP per launch (counted live from this run's
    # plan) / the family's average launch duration in the committed rocprofv3 kernel trace of this same command -- valid only
    # while the kernel sources are the ones that trace was taken from.  The live HIP-event timing of the same launches is
    # reported beside it (`events`): raw pairs over-state a launch by the cost of the pair itself, pairs minus the calibrated
    # empty-pair cost under-state it; when the committed trace is stale the RAW (conservative) event figure is the headline.
    odd_shape = bool(spec.get("shape")) and tuple(spec["shape"][:2]) != DEFAULT_SHAPE
    variant = ("_fp8" if spec["fp8"] else "") + ("_fpn" if spec["fpn"] else "") + ("_r101" if spec["depth"] == 101 else "") + ("_ref600" if odd_shape else "")
    wkey = workload_key(spec["depth"], spec["batch"], spec["proposals"], spec["fp8"], spec["fpn"], spec.get("shape"))
    off, source = offline_profile(dom, variant, wkey)
    ev_raw_us = (f["seconds"] + f["plan_calls"] * f["event_pair_overhead_us"] * 1e-6) / f["launches"] * 1e6
    ev_net_us = f["seconds"] / f["launches"] * 1e6
    if off and off.get("avg_launch_us"):
        head_us, head_src = float(off["avg_launch_us"]), "rocprofv3 kernel trace, " + source
    else:
        head_us, head_src = ev_raw_us, "live HIP events, raw pairs (%s)" % source
    achieved = gflop_per_launch / head_us * 1e3       # GFLOP / us = PFLOP/s
    off_all = {}
    for k in fam:
        off_all[k] = offline_profile(k, variant, wkey)[0]
    families = {}
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1]["seconds"]):
        o = off_all.get(k)
        # durations: the committed trace where it is current (same source as the headline), else raw event pairs
        sec = o["ms_per_step"] * 1e-3 if (o and o.get("ms_per_step")) else v["seconds"] + v["plan_calls"] * v["event_pair_overhead_us"] * 1e-6
        e = {"launches_per_step": v["launches"], "plan_calls_per_step": v["plan_calls"], "ms_per_step": round(sec * 1e3, 4), "timing": "rocprof" if (o and o.get("ms_per_step")) else "events_raw",
             "events_net_ms_per_step": round(v["seconds"] * 1e3, 4)}
        if v["flops"] > 0:
            e["tflops"] = round(v["flops"] / sec / 1e12, 2)
            e["frac_of_mfma_peak"] = round(v["flops"] / sec / 1e12 / (PEAK_FP8_TFLOPS if k == FAM_CONV_F8 else PEAK_BF16_TFLOPS), 4)
        elif v["bytes"] > 0 and sec > 0:
            e["algorithmic_MB_per_step"] = round(v["bytes"] / 1e6, 2)
            e["algorithmic_GBs"] = round(v["bytes"] / sec / 1e9, 1)
            e["frac_of_hbm_peak"] = round(v["bytes"] / sec / 1e9 / PEAK_HBM_GBS, 4)
        if o and o.get("hbm_bytes_per_launch"):
            e["hbm_MB_per_step_pmc"] = round((o.get("hbm_read_bytes_per_step", 0) + o.get("hbm_write_bytes_per_step", 0)) / 1e6, 1)
        for extra in ("roi_bwd_dense_us", "roi_bwd_launches", "roi_bwd_nonzero_fraction_in_step"):
            if extra in v:
                e[extra] = v[extra]
        families[k] = e
    out["roofline"] = {"bound": "mfma", "kernel": dom, "achieved": round(achieved, 2), "peak": peak,
                       "unit": "TFLOP/s", "frac": round(achieved / peak, 5),
                       "traffic": None if not off else off.get("hbm_bytes_per_launch"),
                       "traffic_unit": "HBM bytes per launch (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)",
                       "source": head_src,
                       "launches_per_step": f["launches"], "avg_launch_us": round(head_us, 2),
                       "algorithmic_gflop_per_launch": round(gflop_per_launch, 4),
                       "events": {"avg_launch_us_raw": round(ev_raw_us, 2), "avg_launch_us_minus_empty_pair": round(ev_net_us, 2),
                                  "empty_pair_us": round(f["event_pair_overhead_us"], 2),
                                  "frac_raw": round(gflop_per_launch / ev_raw_us * 1e3 / peak, 5),
                                  "frac_minus_empty_pair": round(gflop_per_launch / ev_net_us * 1e3 / peak, 5)},
                       "whole_step_tflops": round(sum(v["flops"] for v in fam.values()) / (ms * 1e-3) / 1e12, 1),
                       "families": families}


def self_launch(n, argv):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as a CHILD launcher (this process has not touched the GPU and
    never will), relay rank 0's JSON line and exit with the children's status."""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:
        if ln.startswith("{") and '"metric"' in ln:
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    return rc if rc != 0 or line is not None else 4


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-per-gpu", type=int, default=4)
    ap.add_argument("--windows", type=int, default=5, help="time the K-step region this many times back to back (the first is `value`)")
    ap.add_argument("--resident-batches", type=int, default=8, help="synthetic batches resident in HBM that the steps of a region rotate through")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--no-segmented", action="store_true", help="skip the segmented-replay and forced-collectives legs (config.segmented_ms_per_step, config.forced_collectives_world1)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip BASELINE.json configs[3] / configs[4] after the headline (other_configs)")
    ap.add_argument("--profile-steps", type=int, default=3)
    ap.add_argument("--lr-scale", type=float, default=0.01)
    ap.add_argument("--fpn", action="store_true", help="feature-pyramid topology (models/fpn.py): with --fp8 --batch-per-gpu 8 this is BASELINE.json configs[4]")
    ap.add_argument("--fp8", action="store_true", help="fp8 (e4m3) MFMA conv path where a layer supports it (BASELINE.json configs[4]'s precision)")
    ap.add_argument("--depth", type=int, default=50, help="ResNet depth (101 with --proposals 1000 --batch-per-gpu 2 = BASELINE.json configs[3])")
    ap.add_argument("--proposals", type=int, default=0, help="RPN NMS max_total_size / max_output_size_per_class (0: config.json's 300)")
    ap.add_argument("--image-shape", type=int, nargs=2, default=None, metavar=("H", "W"),
                    help="image size (default 375 1242 = BASELINE.json; 600 1987 with --batch-per-gpu 2 = the reference's own config.json: other_configs.reference_default)")
    argv = sys.argv[1:] if argv is None else list(argv)
    args = ap.parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus, argv)
    # stdout carries ONE JSON line and nothing else: everything any library prints there while the benchmark runs (RCCL greets with a
    # version banner on stdout when a communicator is created) is sent to stderr; the line itself goes to the saved descriptor
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    D = importlib.import_module("2d_object_detection_amd.distributed")
    import torch.distributed as dist

    rank, world, local_rank = D.init_from_env()
    assert world == args.gpus, "launched with WORLD_SIZE=%d but --gpus %d" % (world, args.gpus)
    local_rank = int(os.environ.get("FRCNN_BENCH_DEVICE", local_rank))      # (rehearsals: several ranks on one GPU)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    shape = tuple(args.image_shape) if args.image_shape and tuple(args.image_shape) != DEFAULT_SHAPE else None
    spec = {"depth": args.depth, "batch": args.batch_per_gpu, "proposals": args.proposals, "fp8": args.fp8, "fpn": args.fpn, "shape": shape}
    is_headline = spec == BASE_SPEC
    out = measure(spec, args, rank, world, dev, args.windows, not args.no_segmented, not args.no_segmented)
    cfg = out.pop("_cfg")
    loss_vals = out["final_losses"]
    if rank == 0 and world == 1 and is_headline and not args.no_other_configs:
        # the other single-GPU configurations of BASELINE.json, AFTER the headline's timed region: their images/s and roofline get
        # a record from the same run.  3 windows each; a failure here is reported, it does not touch the headline.
        others = {}
        for name, ospec in OTHER_CONFIGS.items():
            try:
                sub = copy.copy(args)
                o = measure(ospec, sub, rank, world, dev, 3, False, False, discard=1)
                o.pop("_cfg")
                keep = {k: o[k] for k in ("value", "unit", "ms_per_step", "dtype", "windows", "final_losses") if k in o}
                keep["workload"] = o["config"]["workload"]
                keep["kernel_launches_per_step"] = o["config"]["kernel_launches_per_step"]
                if "fp8_status" in o["config"]:
                    keep["fp8_status"] = o["config"]["fp8_status"]
                if "roofline" in o:
                    keep["roofline"] = o["roofline"]
                others[name] = keep
            except Exception as e:
                others[name] = {"error": "%s: %s" % (type(e).__name__, e)}
        out["other_configs"] = others
    if rank == 0:
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(cfg)
        bad = [k for k, v in loss_vals.items() if not (v == v and abs(v) != float("inf"))]
        if bad:
            out["error"] = "non-finite losses after the timed window: %s -- the timed workload is degenerate, the number is void" % bad
        if "other_configs" in out:
            # LAST key of the line, compact: a reader who only keeps the tail of stdout still gets every configuration's figures
            out["other_configs_summary"] = {"fields": ["images_per_s", "ms_per_step", "roofline_frac"]}
            for name, o in out["other_configs"].items():
                out["other_configs_summary"][name] = ([o["value"], o["ms_per_step"], (o.get("roofline") or {}).get("frac")]
                                                      if "value" in o else [None, None, o.get("error", "?")[:80]])
        print(json.dumps(out), file=json_out, flush=True)
    if dist.is_initialized():
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()
    if not all(v == v and abs(v) != float("inf") for v in loss_vals.values()):
        return 3
    return 0


if __name__ == "__main__":
    sys.exit(main())
