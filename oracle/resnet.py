"""Keras-applications ResNet50 (v1) truncated at conv4_block6_out, restated in torch-CPU fp32
(test oracle).  reference models/feature_extractor.py:4-11.

[TF-ext] topology per SURVEY.md A.2: ZeroPad(3) -> conv1_conv 7x7/2 valid (bias) -> BN(eps
1.001e-5, momentum .99) -> ReLU -> ZeroPad(1) -> MaxPool 3x3/2 valid -> stacks (64,3,s1),
(128,4,s2), (256,6,s2); the stride sits on the first 1x1 and on the shortcut; all convs have
bias.  PARITY UNPINNED against TensorFlow; torch's own conv/BN/pool are the arithmetic.

Parameters use Keras variable names and Keras layouts (conv kernel HWIO).
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1.001e-5
BN_MOMENTUM = 0.99
STACKS = {50: ((64, 3, 1), (128, 4, 2), (256, 6, 2)), 101: ((64, 3, 1), (128, 4, 2), (256, 23, 2))}
CAFFE_MEAN_BGR = (103.939, 116.779, 123.68)


def conv_specs(depth=50):
    """Ordered list of (name, kh, kw, cin, cout, stride, pad) for every backbone conv."""
    specs = [("conv1", 7, 7, 3, 64, 2, 3)]
    cin = 64
    for si, (f, nblocks, s1) in enumerate(STACKS[depth]):
        stage = si + 2
        for b in range(1, nblocks + 1):
            n = "conv%d_block%d" % (stage, b)
            s = s1 if b == 1 else 1
            if b == 1:
                specs.append((n + "_0", 1, 1, cin, 4 * f, s, 0))
            specs.append((n + "_1", 1, 1, cin, f, s, 0))
            specs.append((n + "_2", 3, 3, f, f, 1, 1))
            specs.append((n + "_3", 1, 1, f, 4 * f, 1, 0))
            cin = 4 * f
    return specs


def param_shapes(depth=50):
    """name -> shape (Keras layouts), insertion-ordered; trainable and BN moving stats."""
    shapes = {}
    for (n, kh, kw, cin, cout, _, _) in conv_specs(depth):
        shapes[n + "_conv/kernel"] = (kh, kw, cin, cout)
        shapes[n + "_conv/bias"] = (cout,)
        shapes[n + "_bn/gamma"] = (cout,)
        shapes[n + "_bn/beta"] = (cout,)
        shapes[n + "_bn/moving_mean"] = (cout,)
        shapes[n + "_bn/moving_variance"] = (cout,)
    return shapes


def init_params(depth=50, seed=0, randomize_affine=False):
    g = torch.Generator().manual_seed(seed)
    p = {}
    for name, shape in param_shapes(depth).items():
        if name.endswith("/kernel"):
            fan_in = shape[0] * shape[1] * shape[2]
            p[name] = torch.randn(shape, generator=g) * math.sqrt(2.0 / fan_in)
        elif name.endswith("/gamma"):
            p[name] = torch.ones(shape) + (0.1 * torch.randn(shape, generator=g) if randomize_affine else 0)
        elif name.endswith("/moving_variance"):
            p[name] = torch.ones(shape)
        elif randomize_affine and (name.endswith("/beta") or name.endswith("/bias")):
            p[name] = 0.1 * torch.randn(shape, generator=g)
        else:
            p[name] = torch.zeros(shape)
    return p


def preprocess(images_u8):
    """uint8 [B,H,W,3] RGB -> fp32 [B,3,H,W] BGR minus caffe mean (feature_extractor.py:6-7)."""
    x = images_u8.to(torch.float32)
    x = x[..., [2, 1, 0]] - torch.tensor(CAFFE_MEAN_BGR, dtype=torch.float32)
    return x.permute(0, 3, 1, 2).contiguous()


def bf16_storage(t):
    """Emulates storing a tensor as bf16 (round-to-nearest-even) with a straight-through gradient.
    The HIP path keeps activations in bf16 between kernels; passing this as `quant=` makes the
    oracle round at exactly the same points (conv output, BN/ReLU output) so that parity compares
    like with like.  On a randomly initialised ResNet in training-mode BN the fp32 and the
    bf16-storage oracles differ by ~23% at conv4_block6_out (error grows ~3% per block; measured,
    see DESIGN.md), so the un-quantised fp32 oracle is not a usable yardstick for a bf16 path."""
    return t + (t.to(torch.bfloat16).to(torch.float32) - t).detach()


def _conv(x, p, n, stride, pad, quant=None):
    w = p[n + "_conv/kernel"].permute(3, 2, 0, 1)
    y = F.conv2d(x, w, p[n + "_conv/bias"], stride=stride, padding=pad)
    return quant(y) if quant is not None else y


def _bn(x, p, n, training, new_stats):
    g, b = p[n + "_bn/gamma"], p[n + "_bn/beta"]
    if training:
        rm = p[n + "_bn/moving_mean"].detach().clone()
        rv = p[n + "_bn/moving_variance"].detach().clone()
        y = F.batch_norm(x, rm, rv, g, b, training=True, momentum=1.0 - BN_MOMENTUM, eps=BN_EPS)
        new_stats[n + "_bn/moving_mean"] = rm
        new_stats[n + "_bn/moving_variance"] = rv
        return y
    return F.batch_norm(x, p[n + "_bn/moving_mean"], p[n + "_bn/moving_variance"], g, b, training=False, eps=BN_EPS)


def forward(p, images_u8, training, depth=50, taps=None, quant=None):
    """Returns (feature_maps NHWC [B,gh,gw,1024], new_bn_stats dict).  `taps` (optional dict)
    receives intermediate NHWC activations for layer-level parity tests.  `quant` (optional,
    e.g. bf16_storage) is applied wherever the HIP path stores an activation."""
    new_stats = {}
    Q = quant if quant is not None else (lambda t: t)
    x = Q(preprocess(images_u8))
    x = _conv(x, p, "conv1", 2, 3, quant)
    x = Q(F.relu(_bn(x, p, "conv1", training, new_stats)))
    if taps is not None:
        taps["conv1_relu"] = x.permute(0, 2, 3, 1)
    x = F.max_pool2d(F.pad(x, (1, 1, 1, 1)), 3, 2)
    if taps is not None:
        taps["pool1_pool"] = x.permute(0, 2, 3, 1)
    for si, (f, nblocks, s1) in enumerate(STACKS[depth]):
        for b in range(1, nblocks + 1):
            n = "conv%d_block%d" % (si + 2, b)
            s = s1 if b == 1 else 1
            if b == 1:
                sc = Q(_bn(_conv(x, p, n + "_0", s, 0, quant), p, n + "_0", training, new_stats))
            else:
                sc = x
            y = Q(F.relu(_bn(_conv(x, p, n + "_1", s, 0, quant), p, n + "_1", training, new_stats)))
            y = Q(F.relu(_bn(_conv(y, p, n + "_2", 1, 1, quant), p, n + "_2", training, new_stats)))
            y = _bn(_conv(y, p, n + "_3", 1, 0, quant), p, n + "_3", training, new_stats)
            x = Q(F.relu(sc + y))
            if taps is not None:
                taps[n + "_out"] = x.permute(0, 2, 3, 1)
                taps[n + "_out_nchw"] = x            # the node every consumer of the block output hangs on (total gradient w.r.t. it)
    return x.permute(0, 2, 3, 1).contiguous(), new_stats
