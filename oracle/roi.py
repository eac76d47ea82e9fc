"""RoI pooling restated from reference models/detectors/fast_rcnn_detector.py:133-177
(test oracle, differentiable torch-CPU fp32).

[TF-ext] tf.image.crop_and_resize (bilinear, extrapolation_value=0) restated from its public
contract, SURVEY.md A.4.  PARITY UNPINNED against TensorFlow.
"""
import torch
import torch.nn.functional as F


def crop_and_resize(image, boxes, box_indices, crop_size):
    """image [B,H,W,C] fp32 NHWC; boxes [n,4] normalised [y1,x1,y2,x2]; box_indices [n] int64;
    crop_size (ch, cw).  Returns [n,ch,cw,C].  Differentiable w.r.t. image only."""
    B, H, W, C = image.shape
    ch, cw = crop_size
    y1, x1, y2, x2 = boxes[:, 0], boxes[:, 1], boxes[:, 2], boxes[:, 3]
    f32 = torch.float32
    Hm1 = torch.tensor(float(H - 1), dtype=f32)
    Wm1 = torch.tensor(float(W - 1), dtype=f32)
    if ch > 1:
        hs = (y2 - y1) * Hm1 / float(ch - 1)
        in_y = (y1 * Hm1)[:, None] + torch.arange(ch, dtype=f32)[None, :] * hs[:, None]
    else:
        in_y = (0.5 * (y1 + y2) * Hm1)[:, None]
    if cw > 1:
        ws = (x2 - x1) * Wm1 / float(cw - 1)
        in_x = (x1 * Wm1)[:, None] + torch.arange(cw, dtype=f32)[None, :] * ws[:, None]
    else:
        in_x = (0.5 * (x1 + x2) * Wm1)[:, None]
    vy = (in_y >= 0) & (in_y <= Hm1)
    vx = (in_x >= 0) & (in_x <= Wm1)
    iy = torch.where(vy, in_y, torch.zeros_like(in_y))
    ix = torch.where(vx, in_x, torch.zeros_like(in_x))
    t = torch.floor(iy)
    b = torch.ceil(iy)
    ly = (iy - t)[:, :, None, None]
    l = torch.floor(ix)
    r = torch.ceil(ix)
    lx = (ix - l)[:, None, :, None]
    t, b, l, r = t.long(), b.long(), l.long(), r.long()
    bi = box_indices.long()[:, None, None]
    tl = image[bi, t[:, :, None], l[:, None, :]]
    tr = image[bi, t[:, :, None], r[:, None, :]]
    bl = image[bi, b[:, :, None], l[:, None, :]]
    br = image[bi, b[:, :, None], r[:, None, :]]
    top = tl + (tr - tl) * lx
    bot = bl + (br - bl) * lx
    out = top + (bot - top) * ly
    valid = (vy[:, :, None] & vx[:, None, :])[..., None]
    return torch.where(valid, out, torch.zeros_like(out))


def roi_pooling(feature_maps, rois, pooled_size=7, kernel_size=2):
    """reference ROIPooling.call (flatten=True, keep_batch_dim=True).
    feature_maps [B,H,W,C]; rois [B,P,4] relative [x1,y1,x2,y2].  Returns [B,P,ps*ps*C]
    flattened in (h, w, c) order."""
    B, P, _ = rois.shape
    r = rois.reshape(-1, 4)[:, [1, 0, 3, 2]]                                # :157-158
    idx = torch.arange(B).repeat_interleave(P)                             # :163
    cs = pooled_size * kernel_size
    crops = crop_and_resize(feature_maps, r, idx, (cs, cs))                # :160-166
    x = crops.permute(0, 3, 1, 2)
    x = F.max_pool2d(x, kernel_size)                                       # :168 (stride = pool size, valid)
    x = x.permute(0, 2, 3, 1).reshape(B, P, -1)                            # :171-175 flatten h,w,c
    return x
