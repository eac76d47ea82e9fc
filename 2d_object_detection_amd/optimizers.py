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

    def apply_plan(self, plan, grad_scale=1.0):
        st = self.store
        for (b, e, l2) in st.decay_ranges():
            plan.add(ops.sgd_momentum, st.w[b:e], st.g[b:e], self.velocity[b:e], st.wb[b:e], e - b, self.momentum, l2, grad_scale,
                     self.iterations, self.boundaries, self.values, self.nb)

    def state_dict(self):
        return {"velocity": self.velocity.cpu(), "iterations": int(self.iterations.item())}

    def load_state_dict(self, sd):
        self.velocity.copy_(sd["velocity"])
        self.iterations.fill_(int(sd["iterations"]))
