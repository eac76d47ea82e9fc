"""Data parallelism: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI).

The reference is single-device; the natural shard is the image.  Each rank runs
the same static train plan on its own images; the only exchange is a SUM all-reduce of the flat
fp32 gradient buffer, cut into the buckets the ParamStore registered in backward-completion order
(heads+RPN, conv4, conv3, conv2+stem).  Bucket i is reduced on a side stream right after backward
segment i, overlapping with the remaining backward segments; the update segment waits for the
last bucket.  xGMI is point-to-point, so buckets are few and large (51 MB fp32 total, 4 buckets)
rather than many small ones.

Loss semantics (so that N ranks x b images == the reference's single batch of N*b images):
classification losses are MEANS over all sampled rows of the global batch -> each rank scales its
classification gradient by 1/world (FasterRCNN(world_size=...)) and the all-reduce SUMs;
regression losses are SUMS over rows (utils/losses.py:40) -> scale 1, SUM.  BatchNorm statistics
stay per replica (documented deviation from a single big batch).
"""
import os

import torch
import torch.distributed as dist


def force_collectives():
    return os.environ.get("FRCNN_FORCE_COLLECTIVES", "0") not in ("", "0")


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def init_from_env(backend=None, timeout_s=None, force=False):
    """Initialise the default process group from torchrun's environment.  Returns (rank, world, local_rank).
    timeout_s: time-out of every collective of the group, barriers included (default: FRCNN_DIST_TIMEOUT_S or 7200).  The
    backend's own default (10 minutes for nccl/RCCL) is shorter than a validation pass of the chief rank can be: the other
    ranks wait in a barrier for it, and a barrier is a collective under the same watchdog.
    force (or FRCNN_FORCE_COLLECTIVES=1): create the group at world 1 as well."""
    import datetime
    if timeout_s is None:
        timeout_s = float(os.environ.get("FRCNN_DIST_TIMEOUT_S", "7200"))
    timeout = datetime.timedelta(seconds=float(timeout_s))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # FRCNN_FORCE_COLLECTIVES=1: a process group (and, through GradientSynchronizer, real collectives on the comm stream) at world 1
    # too -- the one-GPU rehearsal of the RCCL call path: comm stream, ready events and segment graphs interleaved with
    # dist.all_reduce launches, on the backend a multi-GPU run uses
    if (world > 1 or force or force_collectives()) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(_free_port()) if world == 1 else "29500"
        if backend is None:
            # FRCNN_DIST_BACKEND=gloo: rehearsal of the multi-rank path on a box with fewer GPUs than ranks
            backend = os.environ.get("FRCNN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local_rank), timeout=timeout)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=timeout)
    return rank, world, local_rank


class GradientSynchronizer:
    """Bucketed, overlapped all-reduce of a flat gradient buffer.

    `grad` is the flat fp32 gradient tensor, `buckets` a list of (name, begin, end) in the order the
    buckets become final.  Use `after_segment` as the `sync_fn` of FasterRCNN.train_step."""

    def __init__(self, grad, buckets, group=None, force=None, compress=None):
        """force (default: FRCNN_FORCE_COLLECTIVES): issue the all-reduces at world 1 as well (needs an initialised group).
        compress (default: FRCNN_GRAD_COMPRESS; None / "bf16"): "bf16" sends every bucket as bfloat16 -- cast on the comm stream, SUM
        all-reduce of half the bytes (28.3 MB instead of 56.6 MB per step for ResNet-50), the sum written back
        into the fp32 gradient.  Every rank's contribution is rounded to 8 significand bits before the sum, and so is every partial
        sum of the reduction: on two ranks an element of the reduced gradient is within 2^-8 (sum_r |g_r| + |sum_r g_r|) of the fp32
        all-reduce (tests/test_distributed_gloo.py checks exactly that; a ring over W ranks adds one rounding per hop); the momentum
        buffer and the weights stay fp32.  Off by default: the reference sums fp32 gradients."""
        self.grad = grad
        self.buckets = list(buckets)
        compress = os.environ.get("FRCNN_GRAD_COMPRESS", "") if compress is None else compress
        assert compress in ("", "none", "bf16", None), compress
        self.compress = "bf16" if compress == "bf16" else None
        self._half = torch.empty(max(e - b for _, b, e in self.buckets), dtype=torch.bfloat16, device=grad.device) if self.compress else None
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.active = self.world > 1 or (bool(force_collectives() if force is None else force) and dist.is_initialized())
        self.on_gpu = grad.is_cuda
        self.comm_stream = torch.cuda.Stream() if (self.on_gpu and self.active) else None
        self.bytes_per_step = sum(e - b for _, b, e in self.buckets) * (2 if self.compress else grad.element_size())
        self.calls = 0                               # all-reduces issued so far

    def reduce_bucket(self, i):
        if not self.active:
            return
        self.calls += 1
        _, b, e = self.buckets[i]
        view = self.grad[b:e]
        if self.on_gpu:
            ready = torch.cuda.Event()
            ready.record(torch.cuda.current_stream())
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ready)
                self._all_reduce(view)
        else:
            self._all_reduce(view)

    def _all_reduce(self, view):
        if self.compress:
            # (the buckets are reduced one after the other on the comm stream: one staging buffer serves them all)
            half = self._half[:view.numel()]
            half.copy_(view)
            dist.all_reduce(half, op=dist.ReduceOp.SUM, group=self.group)
            view.copy_(half)
        else:
            dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group)

    def note_replay(self, calls):
        """A captured data-parallel step (FasterRCNN, FRCNN_CAPTURE_COLLECTIVES) replayed `calls` all-reduces without going through reduce_bucket."""
        self.calls += calls

    def wait_all(self):
        if self.active and self.on_gpu:
            torch.cuda.current_stream().wait_stream(self.comm_stream)

    def after_segment(self, seg, nseg):
        """Hook for FasterRCNN.train_step: backward segment `seg` has been enqueued; segments 0..nb-1
        complete buckets 0..nb-1, the last segment (update) must see every reduced bucket."""
        if seg < len(self.buckets):
            self.reduce_bucket(seg)
        if seg == nseg - 2:
            for j in range(seg + 1, len(self.buckets)):     # fewer segments than buckets (should not happen)
                self.reduce_bucket(j)
            self.wait_all()


def shard_batch(global_batch, rank, world):
    """Image range [begin, end) of the global batch owned by `rank` (input_pipeline sharding)."""
    per = global_batch // world
    assert per * world == global_batch, "global batch must be divisible by the world size"
    return rank * per, (rank + 1) * per
