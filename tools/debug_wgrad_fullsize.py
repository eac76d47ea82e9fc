"""Debug aid: at 375x1242 batch 4, re-derive the BatchNorm-backward of a few units from the tensors the HIP path stored
(torch-GPU fp64 as a calculator) and compare: partial sums, dgamma / dbeta, dz bits, sum(dz)."""
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
for (bn, k, gname) in (("conv2_block1", 1, "g1"), ("conv2_block1", 2, "g2"), ("conv2_block3", 1, "g1"), ("conv3_block2", 1, "g1"), ("conv4_block6", 2, "g2")):
    u, a = fe.units[bn][k], fe.acts[bn]
    g = a[gname].double()
    bits = ((u.relu_mask[:, :, None] >> torch.arange(8, dtype=torch.uint8, device="cuda")) & 1).reshape(g.shape).bool()
    gm = torch.where(bits, g, torch.zeros_like(g))
    z = u.z.double()
    mean_true = z.mean(0)
    var_true = z.var(0, unbiased=False)
    mean, invstd = u.mean.double(), u.invstd.double()
    xh = (z - mean) * invstd
    S, Sx = gm.sum(0), (gm * xh).sum(0)
    part = u.bwd_partial.double().sum(0)
    gamma = st.weight(u.name + "_bn/gamma").double()
    m = g.shape[0]
    dz_ref = (gamma * invstd * (gm - S / m - xh * Sx / m))
    dz_ref_bf = dz_ref.to(torch.bfloat16)
    dz_part = (gamma * invstd * (gm - part[0] / m - xh * part[1] / m)).to(torch.bfloat16)
    dz = u.dz
    print("%s_%d: M=%d  mean err %.2e  invstd err %.2e | partial sums vs fp64: sum g*m %.2e (abs %.3e of max |S| %.3e), sum g*m*xhat %.2e | "
          "dbeta err %.2e dgamma err %.2e | dz != bf16(fp64 formula): %.4f%% of elements, != formula with HIP's sums: %.4f%% | "
          "max|sum dz| HIP %.3e, fp64-formula->bf16 %.3e, fp64 unrounded %.3e | sum xhat max %.3e" % (
              bn, k, m, float((mean - mean_true).abs().max() / mean_true.abs().max()), float((invstd - 1 / torch.sqrt(var_true + 1.001e-5)).abs().max() / invstd.abs().max()),
              float((part[0] - S).abs().max() / S.abs().max()), float((part[0] - S).abs().max()), float(S.abs().max()),
              float((part[1] - Sx).abs().max() / Sx.abs().max()),
              float((st.grad(u.name + "_bn/beta").double() - S).abs().max() / S.abs().max()), float((st.grad(u.name + "_bn/gamma").double() - Sx).abs().max() / Sx.abs().max()),
              100.0 * float((dz != dz_ref_bf).float().mean()), 100.0 * float((dz != dz_part).float().mean()),
              float(dz.double().sum(0).abs().max()), float(dz_ref_bf.double().sum(0).abs().max()), float(dz_ref.sum(0).abs().max()),
              float(xh.sum(0).abs().max())), flush=True)
