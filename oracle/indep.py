"""ctypes wrapper around oracle/indep.c: the second, independent scalar-loop restatement of tf.image.crop_and_resize (+ its
image gradient) and tf.image.combined_non_max_suppression (TEST INFRASTRUCTURE; see the header of indep.c)."""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_indep.so")
_lib = None


def _load():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "indep.c")):
            subprocess.check_call(["make", "-C", _HERE, "-s"])
        _lib = ctypes.CDLL(_SO)
        for f in ("indep_crop_and_resize", "indep_crop_and_resize_grad_image", "indep_combined_nms"):
            getattr(_lib, f).restype = None
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def crop_and_resize(image, boxes, box_indices, crop_size):
    """image [B,H,W,C] fp32; boxes [n,4] normalised [y1,x1,y2,x2]; box_indices [n]; crop_size (ch, cw) -> [n,ch,cw,C]"""
    img = np.ascontiguousarray(image.detach().numpy(), dtype=np.float32)
    bx = np.ascontiguousarray(boxes.detach().numpy(), dtype=np.float32)
    bi = np.ascontiguousarray(box_indices.numpy(), dtype=np.int32)
    B, H, W, C = img.shape
    ch, cw = crop_size
    out = np.empty((bx.shape[0], ch, cw, C), np.float32)
    _load().indep_crop_and_resize(_p(img), B, H, W, C, _p(bx), _p(bi), bx.shape[0], ch, cw, _p(out))
    return torch.from_numpy(out)


def crop_and_resize_grad_image(grads, image_shape, boxes, box_indices):
    """grads [n,ch,cw,C] -> gradient w.r.t. the image [B,H,W,C]"""
    g = np.ascontiguousarray(grads.detach().numpy(), dtype=np.float32)
    bx = np.ascontiguousarray(boxes.detach().numpy(), dtype=np.float32)
    bi = np.ascontiguousarray(box_indices.numpy(), dtype=np.int32)
    B, H, W, C = image_shape
    out = np.empty((B, H, W, C), np.float32)
    _load().indep_crop_and_resize_grad_image(_p(g), B, H, W, C, _p(bx), _p(bi), bx.shape[0], g.shape[1], g.shape[2], _p(out))
    return torch.from_numpy(out)


def combined_nms(boxes, scores, max_output_size_per_class, max_total_size, iou_threshold, score_threshold):
    boxes = np.ascontiguousarray(boxes.detach().numpy(), dtype=np.float32)
    scores = np.ascontiguousarray(scores.detach().numpy(), dtype=np.float32)
    B, N, q, _ = boxes.shape
    C = scores.shape[2]
    T = int(max_total_size)
    ob, os_ = np.zeros((B, T, 4), np.float32), np.zeros((B, T), np.float32)
    oc, ov = np.zeros((B, T), np.int32), np.zeros((B,), np.int32)
    _load().indep_combined_nms(_p(boxes), _p(scores), B, N, q, C, int(max_output_size_per_class), T, ctypes.c_float(iou_threshold),
                               ctypes.c_float(score_threshold), _p(ob), _p(os_), _p(oc), _p(ov))
    return torch.from_numpy(ob), torch.from_numpy(os_), torch.from_numpy(oc), torch.from_numpy(ov)
