"""The reference's input pipeline (data/input_pipeline.py) without TensorFlow, sharded for data-parallel training.

`InputPipelineCreator(num_classes, image_shape, max_num_objects).create_input_pipeline(filename, batch_size, training)`
returns an iterable of `(images uint8 [B,H,W,3], classes f32 [B,100,C+1], boxes f32 [B,100,4])` batches with the
reference's output contract (:83-130): image decoded and bilinearly resized to `image_shape` then truncated to uint8;
class ids one-hot encoded *shifted by one* over C+1 columns (column 0 = background, never set), boxes divided by the
ORIGINAL image width / height, both clipped / zero padded to `max_num_objects` rows.  Training pipelines repeat forever,
shuffle through a 1024-record buffer and flip image + boxes horizontally with probability 0.5 (:36-38,43-72); evaluation
pipelines make one ordered pass and keep the final partial batch.

Sources: the TFRecord files written by the reference's data/build_tf_records.py (or by data/build_records.py here), a list
of them, or a raw KITTI directory (`image_2/*.png` + `label_2/*.txt`).  Data-parallel rank r of w reads records
r, r+w, r+2w, ... (the north-star's "shards the KITTI input_pipeline across the GPUs"); decode runs in a small thread pool
and batches are assembled on a background thread in a ring of pre-allocated pinned buffers (no HIP call off the training
thread: a host allocation there would invalidate a hipGraph capture in progress) and copied to the device on a side stream.

[TF-ext] `tf.image.resize` semantics restated from the public contract (bilinear, half-pixel centres, no antialias,
float32 arithmetic, result truncated by the uint8 cast); unverified against TensorFlow here (not installed)."""
import io
import os
import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import kitti_classes, tfrecord


def resize_bilinear(image, new_height, new_width):
    """uint8/float [h,w,c] -> float32 [new_height,new_width,c]  (tf.image.resize(image, [new_height, new_width]))."""
    h, w = image.shape[:2]
    img = image.astype(np.float32)
    if (h, w) == (new_height, new_width):
        return img

    def weights(out_size, in_size):
        scale = np.float32(in_size) / np.float32(out_size)
        src = (np.arange(out_size, dtype=np.float32) + np.float32(0.5)) * scale - np.float32(0.5)
        fl = np.floor(src)
        lower = np.maximum(fl, 0).astype(np.int64)
        upper = np.minimum(np.ceil(src), in_size - 1).astype(np.int64)
        return lower, upper, (src - fl).astype(np.float32)

    ylo, yhi, yl = weights(new_height, h)
    xlo, xhi, xl = weights(new_width, w)
    xl = xl[None, :, None]
    top_rows, bot_rows = img[ylo], img[yhi]
    top = top_rows[:, xlo] + (top_rows[:, xhi] - top_rows[:, xlo]) * xl
    bot = bot_rows[:, xlo] + (bot_rows[:, xhi] - bot_rows[:, xlo]) * xl
    return top + (bot - top) * yl[:, None, None]


def pad_or_clip(array, rows):
    """:132-161 for the only use the pipeline makes of it: clip / zero-pad the first dimension to `rows`."""
    out = np.zeros((rows,) + array.shape[1:], dtype=array.dtype)
    n = min(rows, array.shape[0])
    out[:n] = array[:n]
    return out


def kitti_label_features(label_file):
    """The label fields the reference keeps (build_tf_records.py:76-90,108-125): class id and [x_min,y_min,x_max,y_max]."""
    name_to_id = kitti_classes.get_name_to_id_map()
    ids, boxes = [], []
    with open(label_file, "r") as fh:
        for line in fh:
            fields = line.split()
            if fields and fields[0] in name_to_id:
                ids.append(name_to_id[fields[0]])
                boxes.append([float(v) for v in fields[4:8]])
    boxes = np.asarray(boxes, dtype=np.float32).reshape(-1, 4)
    return np.asarray(ids, dtype=np.int64), boxes


def example_from_files(image_file, label_file):
    """Feature dict of one KITTI frame, as data/build_tf_records.py:70-105 stores it."""
    from PIL import Image
    with open(image_file, "rb") as fh:
        encoded = fh.read()
    with Image.open(io.BytesIO(encoded)) as im:
        width, height = im.size
    ids, boxes = kitti_label_features(label_file)
    return {"image/encoded": encoded, "image/width": [width], "image/height": [height], "label/ids": ids,
            "label/x_mins": boxes[:, 0], "label/y_mins": boxes[:, 1], "label/x_maxs": boxes[:, 2], "label/y_maxs": boxes[:, 3]}


class _Dataset:
    """Iterable of batches; `enumerate()` mirrors the tf.data method the reference driver calls (:153)."""

    def __init__(self, creator, sources, batch_size, training, rank, world_size, seed, device, num_workers, prefetch,
                 shuffle_buffer):
        self.c, self.sources, self.batch_size, self.training = creator, sources, batch_size, training
        self.rank, self.world_size, self.seed, self.device = rank, world_size, seed, device
        self.num_workers, self.prefetch, self.shuffle_buffer = num_workers, prefetch, shuffle_buffer

    # -- record stream of this rank (encoded examples or (image, label) file pairs)
    def _records(self):
        i = 0                                            # index of the next record in the concatenation of all sources
        for kind, src in self.sources:
            if kind == "tfrecord":                       # record i of the concatenation belongs to rank i % world
                yield from tfrecord.read_records(src, start=(self.rank - i) % self.world_size, step=self.world_size)
                if len(self.sources) > 1:
                    i += self._scan(src)
            else:
                for pair in src:
                    if i % self.world_size == self.rank:
                        yield pair
                    i += 1

    _counts = {}

    @classmethod
    def _scan(cls, path):
        """Number of records of a TFRecord file (headers only are read)."""
        import struct
        key = (path, os.path.getmtime(path))
        if key not in cls._counts:
            n = 0
            with open(path, "rb") as fh:
                while True:
                    head = fh.read(12)
                    if len(head) < 12:
                        break
                    fh.seek(struct.unpack("<Q", head[:8])[0] + 4, 1)
                    n += 1
            cls._counts[key] = n
        return cls._counts[key]

    def _stream(self, rng):
        if not self.training:
            yield from self._records()
            return
        buf = []                                         # repeat().shuffle(1024): uniform pick from a sliding buffer
        while True:
            empty = True
            for rec in self._records():
                empty = False
                if len(buf) < self.shuffle_buffer:
                    buf.append(rec)
                    continue
                j = int(rng.integers(len(buf)))
                out, buf[j] = buf[j], rec
                yield out
            if empty:
                raise ValueError("input pipeline: rank %d of %d has no records" % (self.rank, self.world_size))
            if len(buf) < self.shuffle_buffer:           # data set smaller than the buffer: one shuffled pass per epoch
                order = rng.permutation(len(buf))
                for j in order:
                    yield buf[j]
                buf = []

    def _batches(self, ring):
        """ring: pre-allocated pinned (images, classes, boxes) buffers to assemble the batches in, or None.  Nothing here calls
        the HIP runtime: this runs on a background thread, and a host allocation there would invalidate a stream capture
        (hipGraph) in progress on the training thread."""
        rng = np.random.default_rng([0x2D0D, self.rank] if self.seed is None else [self.seed, self.rank])
        flips = np.random.default_rng([0xF11B, self.rank] if self.seed is None else [self.seed, self.rank, 1])
        pool = ThreadPoolExecutor(max_workers=max(1, self.num_workers))
        k = 0
        try:
            stream = self._stream(rng)
            while True:
                recs = []
                for rec in stream:
                    recs.append(rec)
                    if len(recs) == self.batch_size:
                        break
                if not recs:
                    return
                do_flip = [bool(flips.random() > 0.5) if self.training else False for _ in recs]
                items = list(pool.map(self.c._decode_and_preprocess, recs, do_flip))
                parts = [torch.from_numpy(np.stack([it[j] for it in items])) for j in range(3)]
                if ring is not None:
                    slot = ring[k % len(ring)]
                    k += 1
                    parts = [buf[:len(recs)].copy_(t) for buf, t in zip(slot, parts)]
                yield tuple(parts)
                if len(recs) < self.batch_size:
                    return
        finally:
            pool.shutdown(wait=False)

    def __iter__(self):
        depth = max(1, self.prefetch)
        q = queue.Queue(maxsize=depth)
        stop = threading.Event()
        END = object()
        on_gpu = self.device is not None and torch.device(self.device).type == "cuda"
        ring, copy_stream = None, None
        if on_gpu:
            # allocated here, on the consumer's thread, before the producer starts; depth + 2 slots: `depth` queued, one being
            # filled, one whose host-to-device copy is being waited for
            h, w, b, c = self.c.image_shape[0], self.c.image_shape[1], self.batch_size, self.c.num_classes
            ring = [(torch.empty(b, h, w, 3, dtype=torch.uint8).pin_memory(),
                     torch.empty(b, self.c.max_num_objects, c + 1).pin_memory(),
                     torch.empty(b, self.c.max_num_objects, 4).pin_memory()) for _ in range(depth + 2)]
            copy_stream = torch.cuda.Stream(device=self.device)

        def produce():
            try:
                for batch in self._batches(ring):
                    while not stop.is_set():
                        try:
                            q.put(batch, timeout=0.1)
                            break
                        except queue.Full:
                            continue
                    if stop.is_set():
                        return
                q.put(END)
            except BaseException as e:                   # surfaced in the consumer
                q.put(e)

        t = threading.Thread(target=produce, daemon=True)
        t.start()
        try:
            while True:
                item = q.get()
                if item is END:
                    return
                if isinstance(item, BaseException):
                    raise item
                if on_gpu:
                    # copy on a side stream and wait for the COPY only (not for the training stream): the pinned slot is free
                    # again when the wait returns
                    with torch.cuda.stream(copy_stream):
                        item = tuple(x.to(self.device, non_blocking=True) for x in item)
                    copy_stream.synchronize()
                    for x in item:
                        x.record_stream(torch.cuda.current_stream(self.device))
                elif self.device is not None:
                    item = tuple(x.to(self.device) for x in item)
                yield item
        finally:
            stop.set()

    def enumerate(self, start=0):
        return enumerate(iter(self), start)


class InputPipelineCreator(object):
    def __init__(self, num_classes, image_shape, max_num_objects=100):
        """data/input_pipeline.py:5-17."""
        self.num_classes = num_classes
        self.image_shape = image_shape
        self.max_num_objects = max_num_objects

    def create_input_pipeline(self, filename, batch_size=1, training=False, rank=0, world_size=1, seed=None, device=None,
                              num_workers=4, prefetch=4, shuffle_buffer=1024):
        """data/input_pipeline.py:19-42.  `filename`: a TFRecord file, a list of them, or a KITTI directory holding
        image_2/ and label_2/.  rank / world_size select this process's shard of the records."""
        names = [filename] if isinstance(filename, (str, os.PathLike)) else list(filename)
        sources = []
        for name in names:
            name = os.fspath(name)
            if os.path.isdir(name):
                img_dir, lab_dir = os.path.join(name, "image_2"), os.path.join(name, "label_2")
                if not (os.path.isdir(img_dir) and os.path.isdir(lab_dir)):
                    raise FileNotFoundError("%s: expected image_2/ and label_2/ (raw KITTI layout)" % name)
                files = sorted(f for f in os.listdir(img_dir) if f.lower().endswith(".png"))
                sources.append(("files", [(os.path.join(img_dir, f), os.path.join(lab_dir, os.path.splitext(f)[0] + ".txt")) for f in files]))
            elif os.path.isfile(name):
                sources.append(("tfrecord", name))
            else:
                raise FileNotFoundError(name)
        if not (0 <= rank < world_size):
            raise ValueError("rank %d outside world of %d" % (rank, world_size))
        return _Dataset(self, sources, batch_size, training, rank, world_size, seed, device, num_workers, prefetch, shuffle_buffer)

    def _augment(self, image, classes, boxes, do_flip):
        """:43-72 with the coin made by the caller (`do_flip = uniform() > 0.5`)."""
        if not do_flip:
            return image, classes, boxes
        flipped = boxes.copy()
        flipped[:, 0] = np.float32(1.0) - boxes[:, 2]
        flipped[:, 2] = np.float32(1.0) - boxes[:, 0]
        return np.ascontiguousarray(image[:, ::-1]), classes, flipped

    def _decode_and_preprocess(self, value, do_flip=False):
        """:83-130.  `value`: a serialized tf.train.Example, or an (image file, label file) pair of a raw KITTI directory."""
        from PIL import Image
        feats = example_from_files(*value) if isinstance(value, tuple) else tfrecord.parse_example(value)
        width, height = int(feats["image/width"][0]), int(feats["image/height"][0])
        encoded = feats["image/encoded"]
        encoded = encoded[0] if isinstance(encoded, list) else encoded
        with Image.open(io.BytesIO(encoded)) as im:
            image = np.asarray(im.convert("RGB"))
        if image.shape != (height, width, 3):
            raise ValueError("image is %s but the record says %dx%d" % (image.shape, height, width))
        new_height, new_width = self.image_shape[0], self.image_shape[1]
        image = resize_bilinear(image, new_height, new_width).astype(np.uint8)           # tf.cast truncates

        ids = np.asarray(feats["label/ids"], dtype=np.int64)
        classes = np.zeros((len(ids), self.num_classes + 1), dtype=np.float32)
        ok = (ids + 1 >= 0) & (ids + 1 <= self.num_classes)                           # tf.one_hot: out-of-range -> zero row
        classes[np.nonzero(ok)[0], ids[ok] + 1] = 1.0
        classes = pad_or_clip(classes, self.max_num_objects)

        w, h = np.float32(width), np.float32(height)
        boxes = np.stack([np.asarray(feats["label/x_mins"], np.float32) / w, np.asarray(feats["label/y_mins"], np.float32) / h,
                          np.asarray(feats["label/x_maxs"], np.float32) / w, np.asarray(feats["label/y_maxs"], np.float32) / h], axis=1)
        boxes = pad_or_clip(boxes.reshape(-1, 4).astype(np.float32), self.max_num_objects)
        # the flip of a padding row would turn [0,0,0,0] into [1,0,1,0] in the reference too (:58-62): kept
        return self._augment(image, classes, boxes, do_flip)
