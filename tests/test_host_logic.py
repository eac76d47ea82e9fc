"""CPU: host-side logic of the product package (no GPU, no kernels launched)."""
import importlib
import json
import os

import torch

RT = importlib.import_module("2d_object_detection_amd.runtime")
OPT = importlib.import_module("2d_object_detection_amd.optimizers")
CFG = importlib.import_module("2d_object_detection_amd.config")
DATA = importlib.import_module("2d_object_detection_amd.data")
DIST = importlib.import_module("2d_object_detection_amd.distributed")


def test_param_store_layout_buckets_and_decay_ranges():
    st = RT.ParamStore(torch.device("cpu"))
    st.register("head/kernel", (64, 10), decay=0.0005)
    st.register("rpn/kernel", (8, 3, 3, 4), decay=0.0005)
    st.register("head/bias", (64,))
    st.end_bucket("heads")
    st.register("conv4/kernel", (16, 1, 1, 8))
    st.register("conv4/gamma", (16,))
    st.end_bucket("conv4")
    st.register("conv1/kernel", (4, 7, 7, 3))
    st.finalize()
    assert [b[0] for b in st.buckets] == ["heads", "conv4", "tail"]
    assert st.buckets[0][1] == 0 and st.buckets[-1][2] == st.size and all(a[2] == b[1] for a, b in zip(st.buckets, st.buckets[1:]))
    assert all(off % 64 == 0 for off, _, _ in st.entries.values())          # 256-byte aligned slices
    r = st.decay_ranges()
    assert len(r) == 2 and r[0][2] == 0.0005 and r[1][2] == 0.0 and r[0][1] == r[1][0] and r[1][1] == st.size
    st.weight("conv4/kernel").fill_(2.0)
    assert float(st.w[st.offset("conv4/kernel")]) == 2.0 and st.grad("conv4/kernel").shape == (16, 1, 1, 8)
    # a second module instance re-attaching to a finalized store must not grow it
    size = st.size
    st.register("conv4/kernel", (16, 1, 1, 8))
    assert st.size == size


def test_plan_segments():
    p = RT.Plan("t")
    log = []
    p.add(log.append, "a")
    p.cut("s1")
    p.add(log.append, "b")
    p.add(log.append, "c")
    p.cut("s2")
    p.cut("s2b")            # empty segment is renamed, not duplicated
    p.add(log.append, "d")
    assert len(p.segments) == 3 and p.segment_names == ["main", "s1", "s2b"] and p.num_launches == 4
    p.run_segment(1)
    assert log == ["b", "c"]
    p.run()
    assert log == ["b", "c", "a", "b", "c", "d"] and not p.captured


def test_plan_branches_are_recorded():
    p = RT.Plan("b")
    log = []
    p.add(log.append, "m0")
    with p.branch("side"):
        p.add(log.append, "s0")
        p.add(log.append, "s1")
    p.add(log.append, "m1")
    p.join("side")
    p.add(log.append, "m2")
    assert [e[3] for e in p.segments[0]] == [None, ("side", False), ("side", False), None, None, None] and p.num_launches == 5
    assert p.segments[0][4][0] is None and p.segments[0][4][1] == ("side",)


def test_piecewise_constant_decay_matches_reference_schedule():
    # train_faster_rcnn.py:62-68,109-112 with Keras' boundary rule (values[i] while step <= boundaries[i]): 1e-3 up to and
    # including step 40000, 1e-4 for 40001..80000, then 1e-5
    s = OPT.PiecewiseConstantDecay([40000, 80000], [1e-3, 1e-4, 1e-5])
    assert s(0) == 1e-3 and s(39999) == 1e-3 and s(40000) == 1e-3 and s(40001) == 1e-4
    assert s(79999) == 1e-4 and s(80000) == 1e-4 and s(80001) == 1e-5
    assert OPT.SGD(0.01).schedule(123) == 0.01


def test_config_schema_matches_reference_keys():
    c = CFG.default_config()
    assert c["image_shape"] == [375, 1242, 3] and c["num_classes"] == 7
    assert set(c) == {"num_classes", "image_shape", "rpn", "rcnn"}
    assert set(c["rpn"]) == {"window_size", "weight_decay", "anchors", "sampling", "nms"}
    assert set(c["rcnn"]) == {"weight_decay", "roi_pooling", "sampling", "nms"}
    for k in ("rpn", "rcnn"):
        assert set(c[k]["sampling"]) == {"foreground_iou_interval", "background_iou_interval", "num_samples", "foreground_proportion"}
        assert set(c[k]["nms"]) == {"score_threshold", "iou_threshold", "max_output_size_per_class", "max_total_size"}
    assert set(c["rpn"]["anchors"]) == {"scales", "aspect_ratios", "base_anchor_shape"}
    assert CFG.default_config((600, 1987, 3))["image_shape"] == [600, 1987, 3]
    json.dumps(c)


def test_synthetic_batch_contract():
    im, gl, gb = DATA.synthetic_batch(3, (375, 1242, 3), seed=7)
    assert im.dtype == torch.uint8 and im.shape == (3, 375, 1242, 3)
    assert gl.shape == (3, 100, 8) and gb.shape == (3, 100, 4)
    real = gl.sum(-1) == 1
    assert bool(((gl.sum(-1) == 0) | real).all()) and float(gl[..., 0].sum()) == 0      # background column never set
    assert bool((real.sum(1) >= 1).all()) and bool((real.sum(1) <= 15).all())
    b = gb[real]
    assert bool((b[:, 2] > b[:, 0]).all() and (b[:, 3] > b[:, 1]).all() and (b >= 0).all() and (b <= 1).all())
    assert float(gb[~real].abs().sum()) == 0
    im2, _, _ = DATA.synthetic_batch(3, (375, 1242, 3), seed=8)
    assert not torch.equal(im, im2)


def test_gradient_synchronizer_single_process_is_a_noop():
    g = torch.arange(10, dtype=torch.float32)
    s = DIST.GradientSynchronizer(g, [("a", 0, 4), ("b", 4, 10)])
    for seg in range(3):
        s.after_segment(seg, 4)
    assert torch.equal(g, torch.arange(10, dtype=torch.float32)) and s.bytes_per_step == 40
    assert DIST.shard_batch(32, 3, 8) == (12, 16)


def test_plan_bucket_cuts_and_sync_points():
    p = RT.Plan("s")
    log = []
    p.add(log.append, "fwd")
    t = torch.ones(3)
    p.sync_point("bn0", [t])                      # a new segment that needs sum-over-ranks(t) first; completes no bucket
    p.add(log.append, "apply")
    p.cut("bwd_conv4")                            # heads bucket done
    p.add(log.append, "conv4")
    p.sync_point("bn1", [t])
    p.add(log.append, "conv4b")
    p.cut("update")                               # conv4 bucket done
    p.add(log.append, "sgd")
    assert p.segment_names == ["main", "bn0", "bwd_conv4", "bn1", "update"]
    assert p.bucket_ends == [1, 3] and sorted(p.pre_sync) == [1, 3]
    p.run_synced()                                # no process group: the collectives are no-ops
    assert log == ["fwd", "apply", "conv4", "conv4b", "sgd"] and torch.equal(t, torch.ones(3))


def _grad_writes(ops, store, fn, args, kwargs):
    """flat-gradient offsets written by one plan launch (tensor arguments that are views of store.g)"""
    base = store.g.untyped_storage().data_ptr()
    tensors = [a for a in list(args) + list(kwargs.values()) if torch.is_tensor(a)]
    if fn is ops.conv2d_wgrad_grouped:
        tensors += [dw for (_d, _x, _dz, dw) in args[0].items]
    return [a.storage_offset() for a in tensors if a.dtype == torch.float32 and a.untyped_storage().data_ptr() == base and a is not store.g
            and a.numel() < store.g.numel()]


def test_gradient_buckets_coincide_with_plan_segments(monkeypatch):
    """No 8-GPU run can catch a mis-cut: on the real train plan (built on CPU tensors, nothing launched) every launch that
    writes into gradient bucket j lies in a segment that ends no later than the j-th bucket cut -- with and without the
    extra segments synchronised BatchNorm inserts -- and the last bucket cut precedes the optimizer update."""
    import copy
    ops = importlib.import_module("2d_object_detection_amd.ops")
    monkeypatch.setattr(ops, "anchors_generate", lambda out, *a, **k: out.zero_())
    monkeypatch.setattr(ops, "clip_to_window", lambda boxes, out, w: out.copy_(boxes))
    M = importlib.import_module("2d_object_detection_amd.models.faster_rcnn")
    cfg = copy.deepcopy(CFG.default_config())
    cfg["image_shape"] = [128, 192, 3]
    for sync_bn in (False, True):
        model = M.FasterRCNN(cfg, device="cpu", world_size=2, sync_bn=sync_bn)
        opt = OPT.SGD(learning_rate=1e-3, momentum=0.9)
        opt.bind(model.store)
        plan = model._build(model._train, 2, True, opt)["plan"]
        buckets = model.store.buckets
        assert [b[0] for b in buckets] == ["heads", "conv4", "conv3", "conv2+stem"]
        assert len(plan.bucket_ends) == len(buckets), (plan.bucket_ends, plan.segment_names)
        assert (len(plan.pre_sync) > 80) == sync_bn
        last_write = [-1] * len(buckets)
        first_update = None
        for si, seg in enumerate(plan.segments):
            for fn, args, kwargs, _br in seg:
                if fn is None:
                    continue
                if fn in (ops.sgd_momentum, ops.sgd_momentum_fused) and first_update is None:
                    first_update = si
                if fn in (ops.sgd_momentum, ops.sgd_momentum_fused):
                    continue
                for off in _grad_writes(ops, model.store, fn, args, kwargs):
                    j = next(i for i, (_n, b, e) in enumerate(buckets) if b <= off < e)
                    last_write[j] = max(last_write[j], si)
        assert all(lw >= 0 for lw in last_write), last_write
        for j, lw in enumerate(last_write):
            assert lw <= plan.bucket_ends[j], "bucket %s is still written in segment %d (%s) after its cut at segment %d" % (
                buckets[j][0], lw, plan.segment_names[lw], plan.bucket_ends[j])
            if j > 0:
                assert lw > plan.bucket_ends[j - 1], "bucket %s is complete before the previous cut: buckets and cuts are out of step" % buckets[j][0]
        assert first_update is not None and first_update > plan.bucket_ends[-1]


def test_early_bucket_updates_follow_every_write_of_their_bucket(monkeypatch):
    """Single GPU, FRCNN_SGD_EARLY=1 (opt-in): the update launches of a train plan tile the flat parameter buffer exactly once, in bucket order; the
    early launch of bucket j comes after the LAST launch that writes a gradient of bucket j (plan order = enqueue order: the trailing side
    stream waits for the main stream's position) and never touches the step counter; the last launch carries stem re-pack and counter.
    With two ranks there is one launch, in the update segment (a bucket's update must follow its all-reduce)."""
    import copy
    ops = importlib.import_module("2d_object_detection_amd.ops")
    monkeypatch.setattr(ops, "anchors_generate", lambda out, *a, **k: out.zero_())
    monkeypatch.setattr(ops, "clip_to_window", lambda boxes, out, w: out.copy_(boxes))
    M = importlib.import_module("2d_object_detection_amd.models.faster_rcnn")
    cfg = copy.deepcopy(CFG.default_config())
    cfg["image_shape"] = [128, 192, 3]
    monkeypatch.setattr(M, "SGD_EARLY", True)         # (opt-in: measured slower on MI355X, models/faster_rcnn.py)
    for topology in ("c4", "fpn"):
        for world in (1, 2):
            model = M.FasterRCNN(cfg, device="cpu", world_size=world, topology=topology)
            opt = OPT.SGD(learning_rate=1e-3, momentum=0.9)
            opt.bind(model.store)
            plan = model._build(model._train, 2, True, opt)["plan"]
            st = model.store
            order, updates, last_write = 0, [], {}
            for si, seg in enumerate(plan.segments):
                for fn, args, kwargs, br in seg:
                    if fn is None:
                        continue
                    order += 1
                    if fn is ops.sgd_momentum_fused:
                        begin = (args[0].data_ptr() - st.w.data_ptr()) // 4
                        updates.append((order, begin, begin + int(args[4]), args[11], br))
                        continue
                    for off in _grad_writes(ops, st, fn, args, kwargs):
                        j = next(i for i, (_n, b, e) in enumerate(st.buckets) if b <= off < e)
                        last_write[j] = order
            if world == 2:
                assert len(updates) == 1 and (updates[0][1], updates[0][2]) == (0, st.size) and updates[0][4] is None
                continue
            assert len(updates) == len(st.buckets), updates
            for j, ((o, b, e, fused, br), (name, bb, be)) in enumerate(zip(updates, st.buckets)):
                assert (b, e) == (bb, be), (name, b, e)
                assert o > last_write[j], "bucket %s is updated at launch %d, but launch %d still writes its gradient" % (name, o, last_write[j])
                is_last = j == len(st.buckets) - 1
                assert bool(fused.arrive) == is_last and (fused.stem_begin >= 0) == is_last
                assert (br is not None and br[0] == "sgd_early" and br[1]) == (not is_last)


def test_bench_self_launches_its_ranks(monkeypatch, capsys):
    """`python bench.py --gpus N` outside torchrun (the way the driver runs N = 1) must not die on WORLD_SIZE != N: the parent starts
    `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD before any GPU call, relays rank 0's JSON line and
    returns the children's status."""
    import io
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    try:
        bench = importlib.import_module("bench")
    finally:
        sys.path.remove(root)
    seen = {}

    class FakeProc:
        def __init__(self, cmd, env=None, stdout=None, text=None):
            seen["cmd"], seen["env"] = cmd, env
            self.stdout = io.StringIO('W0101 torchrun chatter\n{"metric": "m", "value": 1.0, "n_gpus": 2}\n')

        def wait(self):
            return 0

    monkeypatch.setattr(bench.subprocess, "Popen", FakeProc)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench.importlib, "import_module", lambda *a, **k: (_ for _ in ()).throw(AssertionError("parent must not load the GPU stack")))
    rc = bench.main(["--gpus", "2", "--steps", "3", "--warmup", "1"])
    assert rc == 0
    cmd = seen["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "2"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-6:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]
    assert os.path.basename(cmd[cmd.index("--master-port") + 2]) == "bench.py"
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"
    out = capsys.readouterr()
    assert json.loads(out.out.strip())["n_gpus"] == 2 and "torchrun chatter" in out.err


def test_forced_collectives_issue_all_reduces_at_world_one(monkeypatch):
    """FRCNN_FORCE_COLLECTIVES: GradientSynchronizer really calls dist.all_reduce per bucket at world 1 (gloo here; nccl on the GPU)."""
    import torch.distributed as dist
    monkeypatch.setenv("FRCNN_FORCE_COLLECTIVES", "1")
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.delenv("MASTER_PORT", raising=False)
    assert not dist.is_initialized()
    try:
        rank, world, _ = DIST.init_from_env(backend="gloo")
        assert (rank, world) == (0, 1) and dist.is_initialized()
        g = torch.arange(12, dtype=torch.float32)
        sync = DIST.GradientSynchronizer(g, [("a", 0, 4), ("b", 4, 12)])
        assert sync.active
        calls = []
        real = dist.all_reduce
        monkeypatch.setattr(dist, "all_reduce", lambda t, **kw: (calls.append(t.numel()), real(t, **kw))[1])
        sync.after_segment(0, 3)
        sync.after_segment(1, 3)
        assert calls == [4, 8] and sync.calls == 2 and torch.equal(g, torch.arange(12, dtype=torch.float32))
    finally:
        if dist.is_initialized():
            dist.destroy_process_group()
    monkeypatch.setenv("FRCNN_FORCE_COLLECTIVES", "0")
    assert not DIST.GradientSynchronizer(torch.zeros(4), [("a", 0, 4)]).active


def test_nms_threshold_predicate_equals_the_division():
    import numpy as np
    """csrc/boxes_nms.hip::nms_over decides inter / u > thr without the division: above fl(u * fl(thr (1 + 4e-6))) -> True, below
    fl(u * fl(thr (1 - 4e-6))) -> False (both only while the products are normal numbers), the sliver between -> the IEEE division.
    The same fp32 arithmetic in numpy, on quotients packed densely around the threshold (every float within +-64 ulps of thr * u for
    many u) and on random pairs: wherever the kernel skips the division its answer is the division's."""
    rng = np.random.default_rng(0)
    f = np.float32
    for thr in (f(0.7), f(0.5), f(0.6), f(1.0 / 3.0), f(0.05), f(0.95)):
        th_hi, th_lo = f(thr * f(1.0 + 4e-6)), f(thr * f(1.0 - 4e-6))
        u = np.concatenate([rng.uniform(1e-6, 2.0, 4000), 10.0 ** rng.uniform(-36, -28, 500)]).astype(np.float32)
        centre = (thr * u).astype(np.float32)
        steps = np.arange(-64, 65, dtype=np.int32)
        near = (centre.view(np.int32)[:, None] + steps[None, :]).view(np.float32)            # +-64 ulps around fl(thr * u)
        inter = np.concatenate([near.ravel(), (np.repeat(u, 8) * rng.uniform(0, 1, u.size * 8)).astype(np.float32)])
        uu = np.concatenate([np.repeat(u, steps.size), np.repeat(u, 8)])
        hi = (uu * th_hi).astype(np.float32)
        lo = (uu * th_lo).astype(np.float32)
        sure = lo > f(1e-30)
        with np.errstate(all="ignore"):
            exact = (inter / uu).astype(np.float32) > thr
        fast_true = sure & (inter > hi)
        fast_false = sure & ~fast_true & (inter < lo)
        assert np.all(exact[fast_true]) and not np.any(exact[fast_false])
        assert fast_true.sum() + fast_false.sum() > 0.8 * inter.size - near.size      # (the division is the exception, not the rule)


def test_late_zero_fill_precedes_every_use_of_its_buffers(monkeypatch):
    """The backward pass's accumulation targets (flat gradient, stride-2 scatter targets) are zeroed by a launch of the weight-re-layout
    side branch instead of by the fill in front of the step (Plan.zero(late=True) / Plan.late_zero_fill).  On the real train plans (CPU
    tensors, nothing launched): exactly one such launch, in the first segment, inside the branch; no late buffer is also in the
    prologue's list; and every launch that touches a late buffer comes after the JOIN of that branch in plan order."""
    import copy
    ops = importlib.import_module("2d_object_detection_amd.ops")
    monkeypatch.setattr(ops, "anchors_generate", lambda out, *a, **k: out.zero_())
    monkeypatch.setattr(ops, "clip_to_window", lambda boxes, out, w: out.copy_(boxes))
    M = importlib.import_module("2d_object_detection_amd.models.faster_rcnn")
    cfg = copy.deepcopy(CFG.default_config())
    cfg["image_shape"] = [128, 192, 3]
    for topology in ("c4", "fpn"):
        model = M.FasterRCNN(cfg, device="cpu", topology=topology)
        opt = OPT.SGD(learning_rate=1e-3, momentum=0.9)
        opt.bind(model.store)
        plan = (model._build if topology == "c4" else model._build_fpn)(model._train, 2, True, opt)["plan"]
        late = plan._zeros_late
        # (the pyramid's stride-2 scatter targets already hold the neck's gradient when the backbone reaches them: only the flat gradient is late there)
        assert plan._late_placed and len(late) >= (3 if topology == "c4" else 1) and any(t is model.store.g for t in late), (topology, len(late))
        late_ptrs = {t.untyped_storage().data_ptr() for t in late}
        assert not late_ptrs & {t.untyped_storage().data_ptr() for t in plan._zeros}, "a buffer zeroed twice"
        seg0 = plan.segments[0]
        fills = [i for i, e in enumerate(seg0) if e[0] is not None and getattr(e[0], "__name__", "") == "_late_fill"]
        assert len(fills) == 1 and seg0[fills[0]][3] is not None and seg0[fills[0]][3][0] == "weight_flips", fills
        assert not any(getattr(e[0], "__name__", "") == "_late_fill" for s in plan.segments[1:] for e in s if e[0] is not None)
        join = next(i for i, e in enumerate(seg0) if e[0] is None and len(e[1]) == 1 and e[1][0] == "weight_flips")
        assert join > fills[0]

        def touches(args, kwargs):
            found = []
            stack = list(args) + list(kwargs.values())
            while stack:
                a = stack.pop()
                if torch.is_tensor(a):
                    found.append(a)
                elif isinstance(a, (list, tuple)):
                    stack.extend(a)
                elif hasattr(a, "items") and not isinstance(a, dict) and isinstance(getattr(a, "items"), list):   # grouped weight gradients
                    stack.extend(x for it in a.items for x in it)
            return any(t.untyped_storage().data_ptr() in late_ptrs for t in found)

        for i, (fn, args, kwargs, _br) in enumerate(seg0[:join]):
            if fn is None or i == fills[0]:
                continue
            assert not touches(args, kwargs), "%s (entry %d of the first segment) uses a late-zeroed buffer before the branch is joined" % (
                getattr(fn, "__name__", fn), i)


def test_nms_fixed_point_walk_equals_the_greedy_walk():
    """csrc/boxes_nms.hip resolves a team round of <= 512 candidates block by block (64 candidates per block): inside a block the kept set is
    the fixed point of  K <- alive & ~(somebody in K suppresses me)  started from K = alive, later blocks lose the candidates that the
    block's kept set suppresses, and the cap (max_per_class) cuts inside a block.  The same procedure in numpy on random suppression
    relations (`over[i, j]`: candidate i, if kept, suppresses the later candidate j) against the sequential greedy walk."""
    import numpy as np
    rng = np.random.default_rng(5)
    for n, density, cap in ((512, 0.002, 300), (512, 0.05, 300), (256, 0.3, 1000), (512, 0.9, 50), (320, 0.01, 7), (512, 0.0, 300)):
        over = np.triu(rng.random((n, n)) < density, 1)
        valid = rng.random(n) < 0.97
        valid[n - 9:] = False                                   # (the padded tail of a round)
        # greedy, in score order
        kept_g, alive = [], valid.copy()
        for i in range(n):
            if alive[i] and len(kept_g) < cap:
                kept_g.append(i)
                alive[i + 1:] &= ~over[i, i + 1:]
        # block-wise fixed point
        kept_f, alive, rounds_max = [], valid.copy(), 0
        for b0 in range(0, n, 64):
            if len(kept_f) >= cap:
                break
            blk = slice(b0, min(n, b0 + 64))
            al = alive[blk].copy()
            sub = over[blk, blk]                                # sub[i, j]: i suppresses j, both inside the block
            K = al.copy()
            for rounds in range(1, 66):
                Kn = al & ~(sub[K].any(axis=0) if K.any() else np.zeros_like(al))
                if np.array_equal(Kn, K):
                    break
                K = Kn
            assert rounds <= 65
            rounds_max = max(rounds_max, rounds)
            idx = np.flatnonzero(K) + b0
            idx = idx[:cap - len(kept_f)]
            kept_f.extend(idx.tolist())
            alive[blk.stop:] &= ~over[idx, blk.stop:].any(axis=0) if idx.size else True
        assert kept_f == kept_g, (n, density, cap, len(kept_f), len(kept_g))
        assert rounds_max <= 64
