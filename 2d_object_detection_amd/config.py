"""Model configuration: same JSON schema (key names are the API -- the reference splats the
sub-dicts as keyword arguments) and same values as the reference's config.json, with the
BASELINE.json KITTI image size 375x1242 as the default instead of 600x1987."""
import copy
import json

_DEFAULT = {
    "num_classes": 7,
    "image_shape": [375, 1242, 3],
    "rpn": {
        "window_size": 3,
        "weight_decay": 0.0005,
        "anchors": {"scales": [0.25, 0.5, 1.0, 2.0], "aspect_ratios": [0.5, 1.0, 2.0], "base_anchor_shape": [256, 256]},
        "sampling": {"foreground_iou_interval": [0.7, 1.0], "background_iou_interval": [0.0, 0.3],
                     "num_samples": 256, "foreground_proportion": 0.5},
        "nms": {"score_threshold": 0.0, "iou_threshold": 0.7, "max_output_size_per_class": 300, "max_total_size": 300},
    },
    "rcnn": {
        "weight_decay": 0.0005,
        "roi_pooling": {"pooled_size": 7, "kernel_size": 2},
        "sampling": {"foreground_iou_interval": [0.5, 1.0], "background_iou_interval": [0.0, 0.5],
                     "num_samples": 64, "foreground_proportion": 0.25},
        "nms": {"score_threshold": 0.0, "iou_threshold": 0.6, "max_output_size_per_class": 100, "max_total_size": 300},
    },
}


def default_config(image_shape=None):
    c = copy.deepcopy(_DEFAULT)
    if image_shape is not None:
        c["image_shape"] = list(image_shape)
    return c


def load_config(path):
    with open(path) as f:
        return json.load(f)
