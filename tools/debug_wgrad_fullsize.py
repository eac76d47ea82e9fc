"""Debug aid: at 375x1242 batch 4, compare every grouped weight gradient of the train step with an fp64 GEMM of the SAME
bf16 operands (x, dz as the HIP path stored them): isolates the weight-gradient kernels from BatchNorm-backward rounding."""
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import faster_rcnn as O

M = importlib.import_module("2d_object_detection_amd.models.faster_rcnn")
OPT = importlib.import_module("2d_object_detection_amd.optimizers")
cfg = O.default_config((375, 1242, 3))
params = O.init_params(cfg, seed=3, randomize_affine=True)
images, gl, gb = O.synthetic_batch(4, cfg["image_shape"], seed=5)
model = M.FasterRCNN(cfg, sampling_seed=11)
model.use_graphs = False
model.set_weights(params)
model.train_step(images.cuda(), gl.cuda(), gb.cuda(), OPT.SGD(learning_rate=1e-3))
torch.cuda.synchronize()
fe, st = model._train.fe, model.store
x = fe.pool
for (n, ci, f, s, first) in fe.specs:
    u, a = fe.units[n], fe.acts[n]
    for k in sorted(u):
        unit = u[k]
        if unit.k != 1 or unit.stride != 1:
            continue
        xin = {0: x, 1: x, 3: a["a2"]}[k]
        dz = unit.dz.double()
        ref = dz.t() @ xin.double()                                   # [cout, cin]
        got = st.grad(unit.name + "_conv/kernel").view(unit.cout, unit.cin).double()
        err = float((got - ref).norm() / ref.norm())
        sdz = float(unit.dz.float().sum(0).abs().max()), float(unit.dz.float().abs().mean())
        print("%-18s M=%6d %4d->%4d  wgrad vs fp64 GEMM of the same operands: %.2e   |sum dz|max %.3e  mean|dz| %.3e  mean x %.3f std x %.3f" % (
            unit.name, unit.m, unit.cin, unit.cout, err, sdz[0], sdz[1], float(xin.float().mean()), float(xin.float().std())), flush=True)
    x = a["out"]
