"""Synthetic KITTI-like batches with the output contract of the reference's input pipeline
(data/input_pipeline.py:83-130): uint8 images [B,H,W,3]; boxes relative [x_min,y_min,x_max,y_max]
zero-padded to 100 objects; labels one-hot over num_classes+1 columns with the class id shifted by
one (column 0 = background, never set for a real object) and all-zero rows as padding."""
import torch


def synthetic_batch(batch, image_shape, num_classes=7, seed=1234, max_objects=100, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    h, w = image_shape[0], image_shape[1]
    images = torch.randint(0, 256, (batch, h, w, 3), generator=g, dtype=torch.uint8)
    gt_boxes = torch.zeros(batch, max_objects, 4)
    gt_labels = torch.zeros(batch, max_objects, num_classes + 1)
    for b in range(batch):
        n = int(torch.randint(1, 16, (1,), generator=g))
        bw = 0.03 + torch.rand(n, generator=g) * 0.32
        bh = 0.08 + torch.rand(n, generator=g) * 0.52
        x0 = torch.rand(n, generator=g) * (1 - bw)
        y0 = torch.rand(n, generator=g) * (1 - bh)
        gt_boxes[b, :n] = torch.stack([x0, y0, x0 + bw, y0 + bh], 1)
        cls = torch.randint(1, num_classes + 1, (n,), generator=g)
        gt_labels[b, torch.arange(n), cls] = 1.0
    return images.to(device), gt_labels.to(device), gt_boxes.to(device)
