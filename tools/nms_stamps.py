"""Where the proposal / detection NMS kernels of a REAL train step spend their time (kernel-development aid): needs the stamps build
   FRCNN_DEFINES=FRCNN_NMS_STAMPS FRCNN_TAG=nmsst python 2d_object_detection_amd/csrc/build.py
and FRCNN_LIB=lib2dod_hip_nmsst.so.  Prints, per workgroup (image / image x class), the shader clock (units of 100 cycles, about 24 per us) spent in each phase
of nms_class_kernel during the last step.  usage: FRCNN_LIB=lib2dod_hip_nmsst.so python tools/nms_stamps.py [steps]"""
import ctypes
import importlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
L = importlib.import_module("2d_object_detection_amd._lib")
M = importlib.import_module("2d_object_detection_amd.models.faster_rcnn")
OPT = importlib.import_module("2d_object_detection_amd.optimizers")
C = importlib.import_module("2d_object_detection_amd.config")
DATA = importlib.import_module("2d_object_detection_amd.data")

steps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 12
fpn = "fpn" in sys.argv[1:]                     # BASELINE configs[4]: pyramid, fp8, batch 8 (82 k candidates per image)
dev = torch.device("cuda:0")
cfg = C.default_config()
model = M.FasterRCNN(cfg, depth=50, device=dev, seed=0, sampling_seed=0, **(dict(precision="fp8", topology="fpn") if fpn else {}))
opt = OPT.SGD(learning_rate=OPT.PiecewiseConstantDecay([40000, 80000], [1e-4, 1e-5, 1e-6]), momentum=0.9)
batches = [DATA.synthetic_batch(8 if fpn else 4, cfg["image_shape"], seed=1234 + 100 * i, device=dev) for i in range(4)]
for s in range(steps):
    model.train_step(*batches[s % 4], opt)
torch.cuda.synchronize()
lib = ctypes.CDLL(L.LIB_PATH)
buf = (ctypes.c_ulonglong * (2 * 8 * 16))()
rc = lib.frcnn_debug_nms_stamps(buf)
assert rc == 0, rc
names = ["keys", "select", "compact", "sort", "load+kept", "matrix", "walk", "tail"]
for which, title in ((0, "proposal NMS (one workgroup per image)"), (1, "detection NMS (first 8 of image x class workgroups)")):
    print(title)
    for wg in range(8):
        row = [buf[(which * 8 + wg) * 16 + i] for i in range(16)]
        if not any(row):
            continue
        us = [r / 100.0 for r in row[:8]]
        print("  wg %d: " % wg + "  ".join("%s %.1f" % (n, u) for n, u in zip(names, us)) + "  | total %.1f, rounds %d, chunks %d, kept %d | matrix parts: compaction %.1f, operand load %.1f, row loop %.1f | select: %d passes, sweeps %.1f" % (sum(us) + (sum(row[11:14]) + row[15]) / 100.0, row[8], row[9], row[10], row[11] / 100.0, row[12] / 100.0, row[13] / 100.0, row[14], row[15] / 100.0), flush=True)
