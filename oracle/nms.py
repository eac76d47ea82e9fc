"""ctypes wrapper around oracle/nms.c + a slow pure-Python cross-check (test oracle)."""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liboracle_nms.so")
_lib = None


def build():
    if not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(os.path.join(_HERE, "nms.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


def _load():
    global _lib
    if _lib is None:
        _lib = ctypes.CDLL(build())
        _lib.oracle_combined_nms.restype = ctypes.c_int
        _lib.oracle_nms_iou.restype = ctypes.c_float
    return _lib


def combined_nms(boxes, scores, max_output_size_per_class, max_total_size, iou_threshold, score_threshold):
    """boxes [B,N,q,4], scores [B,N,C] (torch fp32).  Returns the 4 outputs of
    tf.image.combined_non_max_suppression (classes as int32)."""
    lib = _load()
    boxes = np.ascontiguousarray(boxes.detach().numpy(), dtype=np.float32)
    scores = np.ascontiguousarray(scores.detach().numpy(), dtype=np.float32)
    B, N, q, _ = boxes.shape
    C = scores.shape[2]
    T = int(max_total_size)
    ob = np.zeros((B, T, 4), np.float32)
    os_ = np.zeros((B, T), np.float32)
    oc = np.zeros((B, T), np.int32)
    ov = np.zeros((B,), np.int32)
    p = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    lib.oracle_combined_nms(p(boxes), p(scores), B, N, q, C, C, int(max_output_size_per_class), T,
                            ctypes.c_float(iou_threshold), ctypes.c_float(score_threshold),
                            p(ob), p(os_), p(oc), p(ov))
    return torch.from_numpy(ob), torch.from_numpy(os_), torch.from_numpy(oc), torch.from_numpy(ov)


def _iou_py(a, b):
    f = np.float32
    y0i, y1i = min(a[0], a[2]), max(a[0], a[2])
    x0i, x1i = min(a[1], a[3]), max(a[1], a[3])
    y0j, y1j = min(b[0], b[2]), max(b[0], b[2])
    x0j, x1j = min(b[1], b[3]), max(b[1], b[3])
    ai = f(f(y1i - y0i) * f(x1i - x0i))
    aj = f(f(y1j - y0j) * f(x1j - x0j))
    if ai <= 0 or aj <= 0:
        return f(0)
    ih = max(f(min(y1i, y1j) - max(y0i, y0j)), f(0))
    iw = max(f(min(x1i, x1j) - max(x0i, x0j)), f(0))
    inter = f(ih * iw)
    return f(inter / f(f(ai + aj) - inter))


def combined_nms_py(boxes, scores, max_output_size_per_class, max_total_size, iou_threshold, score_threshold):
    """Pure-Python loops, small cases only (cross-check of nms.c)."""
    boxes = boxes.numpy().astype(np.float32)
    scores = scores.numpy().astype(np.float32)
    B, N, q, _ = boxes.shape
    C = scores.shape[2]
    T = int(max_total_size)
    ob = np.zeros((B, T, 4), np.float32)
    os_ = np.zeros((B, T), np.float32)
    oc = np.zeros((B, T), np.int32)
    ov = np.zeros((B,), np.int32)
    thr = np.float32(iou_threshold)
    for b in range(B):
        kept = []
        for c in range(C):
            bc = 0 if q == 1 else c
            cand = [(-scores[b, i, c], i) for i in range(N) if scores[b, i, c] > np.float32(score_threshold)]
            cand.sort()
            sel = []
            for negs, i in cand:
                if len(sel) >= max_output_size_per_class:
                    break
                if all(not (_iou_py(boxes[b, i, bc], boxes[b, j, bc]) > thr) for j in sel):
                    sel.append(i)
                    kept.append((negs, len(kept), c, i))
        kept.sort(key=lambda t: (t[0], t[1]))
        kept = kept[:T]
        ov[b] = len(kept)
        for t, (negs, _, c, i) in enumerate(kept):
            ob[b, t] = np.clip(boxes[b, i, 0 if q == 1 else c], 0.0, 1.0)
            os_[b, t] = -negs
            oc[b, t] = c
    return torch.from_numpy(ob), torch.from_numpy(os_), torch.from_numpy(oc), torch.from_numpy(ov)
