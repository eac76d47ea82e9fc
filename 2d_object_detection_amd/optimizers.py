"""SGD with momentum and the piecewise-constant learning-rate schedule of the reference driver
(train_faster_rcnn.py:109-112), evaluated ON THE DEVICE from a device-resident step counter so
that a captured hipGraph keeps following the schedule."""
import torch

from . import ops


class PiecewiseConstantDecay:
    """tf.keras.optimizers.schedules.PiecewiseConstantDecay(boundaries, values) [TF-ext, public contract]: values[0] while
    step <= boundaries[0], values[i] while boundaries[i-1] < step <= boundaries[i], values[-1] for step > boundaries[-1] -- the
    rate changes at the first step AFTER a boundary.  `step` is the optimizer's iteration counter before the update (0 for
    the first update), as Keras passes it (reference train_faster_rcnn.py:62-68,109-112)."""

    def __init__(self, boundaries, values):
        assert len(values) == len(boundaries) + 1
        self.boundaries, self.values = [int(b) for b in boundaries], [float(v) for v in values]

    def __call__(self, step):
        for b, v in zip(self.boundaries, self.values):
            if step <= b:
                return v
        return self.values[-1]


class SGD:
    """tf.keras.optimizers.SGD(learning_rate, momentum): v = m*v - lr*g ; w += v."""

    def __init__(self, learning_rate=0.001, momentum=0.9):
        self.schedule = learning_rate if isinstance(learning_rate, PiecewiseConstantDecay) else PiecewiseConstantDecay([], [float(learning_rate)])
        self.momentum = float(momentum)
        self.store = None

    def bind(self, store):
        if self.store is store:
            return
        assert self.store is None, "optimizer already bound to another model"
        self.store = store
        dev = store.device
        self.velocity = torch.zeros_like(store.w)
        self.iterations = torch.zeros(1, dtype=torch.int64, device=dev)
        nb = len(self.schedule.boundaries)
        self.boundaries = torch.tensor(self.schedule.boundaries + [0], dtype=torch.int64, device=dev)
        self.values = torch.tensor(self.schedule.values, dtype=torch.float32, device=dev)
        self.nb = nb

    def early_ok(self):
        """Can finished gradient buckets be updated one by one (apply_bucket_plan)?  The store's decay ranges must be [regularised | rest]."""
        ranges = self.store.decay_ranges()
        return len(ranges) == 2 and ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] and ranges[1][2] == 0.0

    def apply_bucket_plan(self, plan, begin, end, grad_scale=1.0):
        """The update of ONE finished gradient bucket [begin, end) of the flat buffers, ahead of the plan's last launch (apply_plan with
        the same `first`): the fused kernel over that part, without the stem re-pack and without touching the step counter (the rate is
        read from it: every launch of a step sees the same value, the last one moves it).  Element for element the arithmetic of the
        one-launch form."""
        st = self.store
        ranges = st.decay_ranges()
        decay_end = min(max(ranges[0][1] - begin, 0), end - begin)
        fused = ops.sgd_fused_args(decay_end, ranges[0][2], None)
        plan.hold(fused)
        plan.add(ops.sgd_momentum_fused, st.w[begin:end], st.g[begin:end], self.velocity[begin:end], st.wb[begin:end], end - begin, self.momentum,
                 grad_scale, self.iterations, self.boundaries, self.values, self.nb, fused)

    def apply_plan(self, plan, grad_scale=1.0, stem=None, first=0):
        """The update of a training plan: SGD over the flat buffers, the stem's packed bf16 taps (stem: the backbone's stem unit, or
        None), the step counter + 1 -- ONE launch when the store's decay ranges are [regularised | rest] (they are: the regularised
        kernels are registered first), else one launch per range and the two small ones.  first > 0: the elements below it were
        updated by earlier apply_bucket_plan launches of this plan."""
        st = self.store
        ranges = st.decay_ranges()
        if first > 0:
            assert self.early_ok() and (stem is None or st.offset(stem.name + "_conv/kernel") >= first)
            n = ranges[1][1] - first
            self._arrive = torch.zeros(4, dtype=torch.int32, device=st.device)
            plan.zero(self._arrive)
            decay_end = min(max(ranges[0][1] - first, 0), n)
            if stem is not None:
                fused = ops.sgd_fused_args(decay_end, ranges[0][2], self._arrive, st.offset(stem.name + "_conv/kernel") - first, stem.cout, stem.w_packed)
            else:
                fused = ops.sgd_fused_args(decay_end, ranges[0][2], self._arrive)
            plan.add(ops.sgd_momentum_fused, st.w[first:], st.g[first:], self.velocity[first:], st.wb[first:], n, self.momentum, grad_scale,
                     self.iterations, self.boundaries, self.values, self.nb, fused)
            return
        if len(ranges) == 2 and ranges[0][0] == 0 and ranges[0][1] == ranges[1][0] and ranges[1][2] == 0.0:
            n = ranges[1][1]
            self._arrive = torch.zeros(4, dtype=torch.int32, device=st.device)      # [0]: arrival counter of the fused launch
            plan.zero(self._arrive)                  # (joins the plan's one zero fill: an aborted replay cannot leave it non-zero)
            if stem is not None:
                fused = ops.sgd_fused_args(ranges[0][1], ranges[0][2], self._arrive, st.offset(stem.name + "_conv/kernel"), stem.cout, stem.w_packed)
            else:
                fused = ops.sgd_fused_args(ranges[0][1], ranges[0][2], self._arrive)
            plan.add(ops.sgd_momentum_fused, st.w[:n], st.g[:n], self.velocity[:n], st.wb[:n], n, self.momentum, grad_scale, self.iterations,
                     self.boundaries, self.values, self.nb, fused)
            return
        for (b, e, l2) in ranges:
            plan.add(ops.sgd_momentum, st.w[b:e], st.g[b:e], self.velocity[b:e], st.wb[b:e], e - b, self.momentum, l2, grad_scale,
                     self.iterations, self.boundaries, self.values, self.nb)
        if stem is not None:
            stem.refresh_weights(plan)
        plan.add(ops.step_increment, self.iterations)

    def state_dict(self):
        return {"velocity": self.velocity.cpu(), "iterations": int(self.iterations.item())}

    def load_state_dict(self, sd):
        self.velocity.copy_(sd["velocity"])
        self.iterations.fill_(int(sd["iterations"]))
