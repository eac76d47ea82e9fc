"""Host-side runtime: static launch plans, hipGraph capture and the flat parameter store.

The whole Faster-RCNN step is shape-static (fixed anchors, NMS outputs padded to max_total_size,
fixed sample counts), so it is compiled ONCE into a `Plan`: an ordered list of C-ABI kernel
launches over pre-allocated HBM buffers.  A plan can be replayed eagerly (tests, profiling) or
captured into hipGraphs (training: no per-kernel host cost, no host sync anywhere in the step).
Plans are cut into *segments* so that a data-parallel driver can interleave RCCL all-reduces of
finished gradient buckets with the remaining backward segments.
"""
import os

import numpy as np
import torch


_NOBRANCH = set(filter(None, os.environ.get("FRCNN_NOBRANCH", "").split(",")))   # measuring aid (tools/ab_plan.sh): these branches stay on the main stream
_SERIAL = bool(os.environ.get("FRCNN_SERIAL_PLAN"))      # debugging aid: run branch launches on the main stream
# TIMING-ONLY what-if aid (tools/ab_lib.sh): launches of these ops (function names, comma separated) are dropped from every plan -- the results
# are garbage; the step time says what the family costs ON the critical path, i.e. the most any fusion / overlap of it could save
_SKIP_OPS = set(filter(None, os.environ.get("FRCNN_SKIP_OPS", "").split(",")))


class _Branch:
    def __init__(self, plan, name, follow):
        self.plan, self.name, self.follow = plan, name, follow

    def __enter__(self):
        assert self.plan._branch is None, "plan branches do not nest"
        self.plan._branch = (self.name, self.follow)
        return self

    def __exit__(self, *exc):
        self.plan._branch = None
        return False


_MARK = "mark"            # second element of a Plan.mark entry's args


class Plan:
    """Ordered launches, cut into segments.  Launches added inside `with plan.branch(name):` run on a side stream that
    forks from the main stream at the branch's first launch and is joined back by `plan.join(name)` (or at the end of
    the segment): small latency-bound kernels (NMS, target assignment, weight re-layouts) overlap the convolutions of
    the main stream instead of serialising with them.  Under graph capture the fork / join events become graph edges."""

    def __init__(self, name="plan"):
        self.name = name
        self.segments = [[]]
        self.segment_names = ["main"]
        self.keep = []            # buffers owned by the plan
        self._graphs = None
        self._whole = None
        self._branch = None
        self._streams = {}
        self._zeros, self._zero_built, self._zero_table = [], False, None
        self._zeros_late, self._late_placed, self._late_table = [], False, None
        self.bucket_ends = []     # indices of the segments whose end completes the next gradient bucket (in bucket order)
        self.pre_sync = {}        # segment index -> tensors that must be SUM all-reduced over the ranks before it runs

    # -- construction
    def add(self, fn, *args, **kwargs):
        if _SKIP_OPS and getattr(fn, "__name__", "") in _SKIP_OPS:
            return
        br = None if _SERIAL else self._branch
        if br is not None and br[0] in _NOBRANCH:
            br = None
        self.segments[-1].append((fn, args, kwargs, br))

    def branch(self, name, follow=False):
        """follow=True: the block's first launch additionally waits for everything enqueued on the main stream so far, even
        if the branch already exists (a side stream that trails the main chain: every weight-gradient kernel depends on the
        BN backward kernel just before it, nothing on the main chain depends on the weight gradients)."""
        return _Branch(self, name, follow)

    def join(self, name):
        self.segments[-1].append((None, (name,), {}, None))

    def mark(self, name):
        """The fork point of branch `name`, ahead of its block: the branch's first launch waits for the main stream as of HERE, but the
        block itself may be added (and so enqueued / captured) after the main-stream launches that follow the mark.  Same
        dependencies as a block placed at the mark; a different order of node creation under graph capture (which of a fork's two
        successors the runtime keeps on the predecessor's hardware queue depends on it)."""
        self.segments[-1].append((None, (name, _MARK), {}, None))      # (fn None, like a join: entry walkers skip it)

    def cut(self, name, bucket=True):
        """Start a new segment.  bucket=True: the segment that ends here completes the next gradient bucket (the data-parallel
        driver all-reduces it while the following segments run)."""
        assert self._branch is None
        if self.segments[-1]:
            if bucket:
                self.bucket_ends.append(len(self.segments) - 1)
            self.segments.append([])
            self.segment_names.append(name)
        else:
            self.segment_names[-1] = name

    def sync_point(self, name, tensors):
        """Everything enqueued so far produced per-rank partial sums in `tensors`; what follows needs their SUM over all ranks
        (synchronised BatchNorm statistics).  Starts a new segment whose runner all-reduces `tensors` first (a no-op on one
        rank); collectives are not captured into the segment graphs."""
        self.cut(name, bucket=False)
        self.pre_sync.setdefault(len(self.segments) - 1, []).extend(tensors)

    def zero(self, tensor, late=False):
        """`tensor` must be all zeros when the plan starts (an accumulation target: atomics, scatter, split-K).  All such
        buffers of a plan are zeroed by ONE fill launch at the head of its first segment (frcnn_fill_zero_multi); none of them
        carries state from one run of the plan to the next.  late=True: the buffer is first touched after the point where the
        branch holding the plan's late_zero_fill() entry is joined (the flat gradient, the scatter targets of the backward pass:
        150 MB of the 160 a train step zeroes) -- its fill is that entry's launch, beside the forward pass instead of in front of
        it; a plan without such an entry zeroes it with the others."""
        assert not self._zero_built, "plan already ran: its zero table is frozen"
        (self._zeros_late if late else self._zeros).append(tensor)

    def late_zero_fill(self):
        """Place the fill of the late=True buffers here (inside a side branch of the first segment: everything that accumulates into
        them must come after that branch's join)."""
        assert not self._late_placed and len(self.segments) == 1, "one late fill, in the plan's first segment"
        self._late_placed = True
        self.add(self._late_fill)

    def _late_fill(self):
        if not self._zeros_late:
            return
        from . import ops
        if self._late_table is None:
            self._late_table = ops.make_zero_table(self._zeros_late, self._zeros_late[0].device)
        ops.fill_zero_multi(*self._late_table)

    def _zero_prologue(self):
        if not self._zero_built:
            if not self._late_placed:
                self._zeros, self._zeros_late = self._zeros + self._zeros_late, []
            self._zero_built = True
            if self._zeros:
                from . import ops
                self._zero_table = ops.make_zero_table(self._zeros, self._zeros[0].device)
        if self._zero_table is not None:
            from . import ops
            ops.fill_zero_multi(*self._zero_table)

    def hold(self, *tensors):
        self.keep.extend(tensors)
        return tensors[0] if len(tensors) == 1 else tensors

    @property
    def num_launches(self):
        return sum(1 for s in self.segments for e in s if e[0] is not None) + (1 if self._zeros or (self._zeros_late and not self._late_placed) else 0)

    # -- execution
    def _join(self, main, side, name):
        ev = torch.cuda.Event()
        ev.record(side.pop(name))
        main.wait_event(ev)

    def run_segment(self, i, carry=None):
        """carry (Plan.run): the side streams still open from earlier segments; they are NOT joined at this segment's end -- a branch
        opened with follow=True may trail the main chain across cuts (the early optimizer updates of finished gradient buckets) until
        a join entry or the plan's end.  Stand-alone (segment graphs of a data-parallel step): every branch is joined at the end."""
        main = None                                # (no CUDA call for plans without branches: host-logic tests run on CPU)
        if i == 0:
            self._zero_prologue()
        side = {} if carry is None else carry
        prev_branch = None
        marks = {}
        for fn, args, kwargs, br in self.segments[i]:
            if fn is None and len(args) == 2:
                if not _SERIAL:
                    if main is None:
                        main = torch.cuda.current_stream()
                    marks[args[0]] = torch.cuda.Event()
                    marks[args[0]].record(main)
                continue
            if fn is None:
                if args[0] in side:
                    if main is None:
                        main = torch.cuda.current_stream()
                    self._join(main, side, args[0])
            elif br is None:
                fn(*args, **kwargs)
            else:
                name, follow = br
                if main is None:
                    main = torch.cuda.current_stream()
                first_of_block = prev_branch != br
                if name not in side or (follow and first_of_block):
                    if name not in self._streams:
                        self._streams[name] = torch.cuda.Stream()
                    ev = marks.pop(name, None)
                    if ev is None:
                        ev = torch.cuda.Event()
                        ev.record(main)
                    self._streams[name].wait_event(ev)
                    side[name] = self._streams[name]
                with torch.cuda.stream(side[name]):
                    fn(*args, **kwargs)
            prev_branch = br if fn is not None else prev_branch
        if carry is None:
            for name in list(side):
                self._join(main, side, name)

    def run(self):
        side = {}
        for i in range(len(self.segments)):
            self.run_segment(i, carry=side)
        if side:
            main = torch.cuda.current_stream()
            for name in list(side):
                self._join(main, side, name)

    def sync_before(self, i):
        """All-reduce (SUM over the default process group) the partial sums segment i waits for (sync_point)."""
        if i in self.pre_sync:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
                for t in self.pre_sync[i]:
                    dist.all_reduce(t, op=dist.ReduceOp.SUM)

    def run_synced(self):
        """Eager run including the collectives of the plan's sync points (synchronised BatchNorm on several ranks)."""
        for i in range(len(self.segments)):
            self.sync_before(i)
            self.run_segment(i)

    def capture(self):
        """Capture every segment into its own hipGraph (after one eager warm-up run)."""
        graphs = []
        pool = None
        for i in range(len(self.segments)):
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, pool=pool):
                self.run_segment(i)
            pool = g.pool()
            graphs.append(g)
        self._graphs = graphs
        # the whole step as ONE graph for callers that do not interleave anything between segments (single GPU)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, pool=pool):
            self.run()
        self._whole = g
        return graphs

    def capture_with_hook(self, hook, error_mode="thread_local"):
        """The whole step as ONE hipGraph INCLUDING what a data-parallel driver does between the segments: hook(bucket, n) is called
        inside the capture after every bucket-completing segment (its collectives -- RCCL all-reduces on a side stream that forks
        from / joins the capturing stream -- become nodes of the graph), and the sync points' all-reduces (synchronised BatchNorm)
        are captured on the main stream.  A data-parallel step then is one replay: no event record / stream wait / collective launch
        from the host between segment graphs.  error_mode "thread_local": the process group's watchdog thread may query events while
        this thread captures."""
        g = torch.cuda.CUDAGraph()
        pool = self._graphs[0].pool() if self._graphs else None
        with torch.cuda.graph(g, pool=pool, capture_error_mode=error_mode):
            done = 0
            for i in range(len(self.segments)):
                self.sync_before(i)
                self.run_segment(i)
                if done < len(self.bucket_ends) and i == self.bucket_ends[done]:
                    hook(done, len(self.bucket_ends) + 1)
                    done += 1
        return g

    def replay_segment(self, i):
        self._graphs[i].replay()

    def replay(self):
        self._whole.replay()

    @property
    def captured(self):
        return self._graphs is not None


class ParamStore:
    """All trainable parameters in ONE flat fp32 buffer (+ gradient, momentum and bf16 working
    copy of identical layout).  Registration order is the order gradients become final during
    backward (heads first, stem last), so contiguous slices are all-reduce buckets; parameters
    carrying the L2 kernel regulariser are registered first (one contiguous SGD range)."""

    def __init__(self, device):
        self.device = device
        self.entries = {}          # name -> (offset, shape, decay)
        self.order = []
        self.size = 0
        self.finalized = False
        self.buckets = []          # (name, begin, end)
        self._bucket_start = 0
        self.stats = {}            # non-trainable fp32 tensors (BN moving statistics)

    def register(self, name, shape, decay=0.0):
        if name in self.entries:                  # a second module instance (eval twin) re-attaching to the same store
            assert self.entries[name][1] == tuple(shape), name
            return name
        assert not self.finalized, "ParamStore already finalized: cannot add %s" % name
        n = int(np.prod(shape))
        self.entries[name] = (self.size, tuple(shape), float(decay))
        self.order.append(name)
        self.size += (n + 63) // 64 * 64          # 256-byte aligned slices
        return name

    def end_bucket(self, name):
        if self.size > self._bucket_start:
            self.buckets.append((name, self._bucket_start, self.size))
            self._bucket_start = self.size

    def register_stat(self, name, shape, fill):
        if name in self.stats:
            return self.stats[name]
        self.stats[name] = torch.full(tuple(shape), float(fill), dtype=torch.float32, device=self.device)
        return self.stats[name]

    def finalize(self):
        if self.finalized:
            return
        self.end_bucket("tail")
        self.w = torch.zeros(self.size, dtype=torch.float32, device=self.device)
        self.g = torch.zeros(self.size, dtype=torch.float32, device=self.device)
        self.wb = torch.zeros(self.size, dtype=torch.bfloat16, device=self.device)
        self.finalized = True

    def _view(self, buf, name):
        off, shape, _ = self.entries[name]
        return buf[off:off + int(np.prod(shape))].view(shape)

    def weight(self, name):
        return self._view(self.w, name)

    def grad(self, name):
        return self._view(self.g, name)

    def weight_bf16(self, name):
        return self._view(self.wb, name)

    def offset(self, name):
        return self.entries[name][0]

    def decay_ranges(self):
        """Contiguous (begin, end, l2) ranges covering the whole buffer."""
        ranges = []
        for name in self.order:
            off, shape, decay = self.entries[name]
            end = off + (int(np.prod(shape)) + 63) // 64 * 64
            if ranges and ranges[-1][2] == decay and ranges[-1][1] == off:
                ranges[-1] = (ranges[-1][0], end, decay)
            else:
                ranges.append((off, end, decay))
        return ranges

    def refresh_bf16(self):
        from . import ops
        ops.cast_f32_bf16(self.w, self.wb)


def dev_tensor(shape, dtype, device, fill=None):
    if fill is None:
        return torch.empty(shape, dtype=dtype, device=device)
    return torch.full(shape, fill, dtype=dtype, device=device)
