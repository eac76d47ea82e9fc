"""Boxes drawn on images for the TensorBoard image summaries of the training driver (reference utils/images.py:11-105,
train_faster_rcnn.py:170-195).  Host-side, PIL only: the reference's seaborn "hls" palette is restated from its public definition
([TF-ext]-style: seaborn.hls_palette(n, h=.01, l=.6, s=.65) = n evenly spaced hues through colorsys.hls_to_rgb), and its
tf.io.decode_image round trip is replaced by the PNG bytes themselves -- what the event file stores."""
import colorsys
import io
import math

from PIL import ImageDraw, ImageFont


def hls_palette(n, h=0.01, l=0.6, s=0.65):
    """seaborn.color_palette("hls", n) as 0..255 RGB tuples (reference utils/images.py:64-65)."""
    hues = [((i / float(n)) + h) % 1.0 for i in range(int(n))]
    return [tuple(int(c * 255) for c in colorsys.hls_to_rgb(hue, l, s)) for hue in hues]


def _text_size(font, text):
    if hasattr(font, "getbbox"):                # Pillow >= 8 (getsize, which the reference calls, is gone from Pillow 10)
        x0, y0, x1, y1 = font.getbbox(text)
        return x1 - x0, y1
    return font.getsize(text)


def draw_box_on_image(image, box, label=None, relative=False, color="red", thickness=2):
    """reference utils/images.py:11-45: the box outline as a closed poly-line, the label on a filled rectangle above its top-left
    corner.  box: [x_min, y_min, x_max, y_max], relative to the image size when `relative`."""
    x_min, y_min, x_max, y_max = [float(v) for v in box]
    if relative:
        width, height = image.size
        x_min, x_max, y_min, y_max = x_min * width, x_max * width, y_min * height, y_max * height
    draw = ImageDraw.Draw(image)
    draw.line([(x_min, y_min), (x_max, y_min), (x_max, y_max), (x_min, y_max), (x_min, y_min)], width=thickness, fill=color)
    if label is not None:
        font = ImageFont.load_default()
        text_width, text_height = _text_size(font, label)
        draw.rectangle([(x_min - math.floor(thickness / 2), y_min - text_height), (x_min + text_width + thickness, y_min)], fill=color)
        draw.text((x_min + math.ceil(thickness / 2), y_min - text_height), label, fill="black", font=font)


def draw_predictions_on_image(image, boxes, scores=None, class_indices=None, class_names=None, relative=False, default_color="red",
                              thickness=2):
    """reference utils/images.py:48-83: every box with "<class>: <score>%" in its class colour (or default_color without classes).
    boxes [num, 4], scores [num], class_indices [num]: sequences, numpy arrays or tensors."""
    palette = hls_palette(len(class_names)) if (class_names is not None and class_indices is not None) else None
    for i, box in enumerate(boxes):
        label, color = None, default_color
        if palette is not None:
            ci = int(class_indices[i])
            label, color = class_names[ci], palette[ci]
        if scores is not None:
            label = "" if label is None else label + ": "
            label += "{:.0f}%".format(float(scores[i]) * 100)
        draw_box_on_image(image=image, box=[float(v) for v in box], label=label, relative=relative, color=color, thickness=thickness)


def to_png(image):
    """PNG bytes of a PIL image (the reference's to_tensor decodes them again for tf.summary.image, which re-encodes: utils/images.py:86-105)."""
    buf = io.BytesIO()
    image.save(buf, format="PNG")
    return buf.getvalue()
