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
import collections
import io
import os
import queue
import threading
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import kitti_classes, tfrecord


def resize_bilinear(image, new_height, new_width):
    """uint8/float [h,w,c] -> float32 [new_height,new_width,c]  (tf.image.resize(image, [new_height, new_width])).
    Per output pixel: top = tl + (tr - tl) * x_lerp, bottom = bl + (br - bl) * x_lerp, out = top + (bottom - top) * y_lerp in float32.
    The x interpolation runs once per SOURCE row (two column gathers, made in the source dtype: a quarter of the bytes for uint8) and the
    y interpolation picks its two rows from that: the same operands in the same order as interpolating the four neighbours of every
    output pixel (tests/test_data_metrics.py::test_resize_bilinear_equals_the_four_neighbour_form), at less than half the cost."""
    h, w = image.shape[:2]
    if (h, w) == (new_height, new_width):
        return image.astype(np.float32)

    def weights(out_size, in_size):
        scale = np.float32(in_size) / np.float32(out_size)
        src = (np.arange(out_size, dtype=np.float32) + np.float32(0.5)) * scale - np.float32(0.5)
        fl = np.floor(src)
        lower = np.maximum(fl, 0).astype(np.int64)
        upper = np.minimum(np.ceil(src), in_size - 1).astype(np.int64)
        return lower, upper, (src - fl).astype(np.float32)

    ylo, yhi, yl = weights(new_height, h)
    xlo, xhi, xl = weights(new_width, w)
    left = np.take(image, xlo, axis=1).astype(np.float32)
    right = np.take(image, xhi, axis=1).astype(np.float32)
    right -= left
    right *= xl[None, :, None]
    left += right                                    # [h, new_width, c]: every source row interpolated along x
    top, bot = np.take(left, ylo, axis=0), np.take(left, yhi, axis=0)
    bot -= top
    bot *= yl[:, None, None]
    top += bot
    return top


def resize_to_uint8(image, new_height, new_width):
    """tf.cast(tf.image.resize(image, size), tf.uint8) (:116-117): truncation.  A uint8 image of the target size is returned as it is
    (uint8 -> float32 -> uint8 is the identity)."""
    if image.dtype == np.uint8 and image.shape[:2] == (new_height, new_width):
        return image
    return resize_bilinear(image, new_height, new_width).astype(np.uint8)


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


def example_from_files(image_file, label_file, image_format="keep"):
    """Feature dict of one KITTI frame, as data/build_tf_records.py:70-105 stores it.
    image_format "keep": the file's bytes as they are (the reference tool); "bmp": the frame re-encoded without compression.  A PNG
    frame costs 17-19 ms of inflate on a host core when it is read back, a BMP frame 1 ms, and both are what the reference's
    tf.io.decode_image (data/input_pipeline.py:112) and this pipeline accept: same pixels, 1.7 x the bytes."""
    from PIL import Image
    with open(image_file, "rb") as fh:
        encoded = fh.read()
    with Image.open(io.BytesIO(encoded)) as im:
        width, height = im.size
        if image_format == "bmp":
            out = io.BytesIO()
            im.convert("RGB").save(out, format="BMP")
            encoded = out.getvalue()
        elif image_format != "keep":
            raise ValueError("image_format: 'keep' or 'bmp'")
    ids, boxes = kitti_label_features(label_file)
    return {"image/encoded": encoded, "image/width": [width], "image/height": [height], "label/ids": ids,
            "label/x_mins": boxes[:, 0], "label/y_mins": boxes[:, 1], "label/x_maxs": boxes[:, 2], "label/y_maxs": boxes[:, 3]}


_PIXEL = np.dtype((np.void, 3))


def _place_image(dst, image):
    """dst[...] = image for uint8 [H,W,3]; a horizontally reversed view is copied pixel by pixel (3-byte items: 0.9 ms for a KITTI frame)
    instead of byte by byte (2.3 ms)."""
    if image.dtype == np.uint8 and image.ndim == 3 and image.shape[2] == 3 and image.strides[1] < 0 and dst.flags.c_contiguous:
        base = image[:, ::-1]
        if base.flags.c_contiguous and dst.dtype == np.uint8 and dst.shape == image.shape:
            h, w = image.shape[:2]
            dst.view(_PIXEL).reshape(h, w)[:] = base.view(_PIXEL).reshape(h, w)[:, ::-1]
            return
    dst[...] = image


class _Dataset:
    """Iterable of batches; `enumerate()` mirrors the tf.data method the reference driver calls (:153).

    Throughput (tools/driver_rate.py): one KITTI frame costs 17-19 ms of PNG decoding and, when its size is not the configured one,
    19 ms of resizing on a host core; the train step takes images at 1 ms each.  So (a) decoding runs ahead of the consumer in a window
    of 2 x num_workers records ACROSS batch boundaries (the pool never drains between batches), and (b) the decoded, resized,
    un-flipped records are kept in host memory up to `cache_bytes` (all of KITTI at 375 x 1242 is 10.4 GB): from the second epoch
    on a batch is a flip and a copy into the pinned ring.  Neither changes what the pipeline yields."""

    def __init__(self, creator, sources, batch_size, training, rank, world_size, seed, device, num_workers, prefetch,
                 shuffle_buffer, cache_bytes=0):
        self.c, self.sources, self.batch_size, self.training = creator, sources, batch_size, training
        self.rank, self.world_size, self.seed, self.device = rank, world_size, seed, device
        self.num_workers, self.prefetch, self.shuffle_buffer = num_workers, prefetch, shuffle_buffer
        self.cache_bytes = int(cache_bytes)
        self._cache, self._cached_bytes, self._cache_lock = {}, 0, threading.Lock()
        self.decoded = 0                                 # records decoded so far (cache misses): for tests and tools

    # -- record stream of this rank: (key, encoded example or (image, label) file pair); the key names the record across epochs
    def _records(self):
        i = 0                                            # index of the next record in the concatenation of all sources
        for n, (kind, src) in enumerate(self.sources):
            if kind == "tfrecord":                       # record i of the concatenation belongs to rank i % world
                for j, rec in enumerate(tfrecord.read_records(src, start=(self.rank - i) % self.world_size, step=self.world_size)):
                    yield (n, j), rec
                if len(self.sources) > 1:
                    i += self._scan(src)
            else:
                for j, pair in enumerate(src):
                    if i % self.world_size == self.rank:
                        yield (n, j), pair
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

    def _decode(self, key, value):
        """The decoded, resized, un-flipped record (worker threads); kept for the next epochs while the budget lasts."""
        item = self.c._decode(value)
        with self._cache_lock:
            self.decoded += 1
            size = sum(a.nbytes for a in item)
            if self._cached_bytes + size <= self.cache_bytes and key not in self._cache:
                self._cache[key] = item
                self._cached_bytes += size
        return item

    def _batches(self, ring):
        """ring: pre-allocated pinned (images, classes, boxes) buffers to assemble the batches in, or None.  Nothing here calls
        the HIP runtime: this runs on a background thread, and a host allocation there would invalidate a stream capture
        (hipGraph) in progress on the training thread."""
        rng = np.random.default_rng([0x2D0D, self.rank] if self.seed is None else [self.seed, self.rank])
        flips = np.random.default_rng([0xF11B, self.rank] if self.seed is None else [self.seed, self.rank, 1])
        workers = max(1, self.num_workers)
        pool = ThreadPoolExecutor(max_workers=workers)
        ring_np = None if ring is None else [tuple(t.numpy() for t in slot) for slot in ring]     # (views of the pinned memory)
        window = collections.deque()                     # (key, decoded record or its future, flip coin) in stream order
        pending = {}                                     # key -> future of a record being decoded
        depth = max(self.batch_size, 2 * workers)
        k = 0
        try:
            stream = self._stream(rng)
            more = True
            while True:
                while more and len(window) < depth:      # decode ahead, across batch boundaries
                    try:
                        key, rec = next(stream)
                    except StopIteration:
                        more = False
                        break
                    coin = bool(flips.random() > 0.5) if self.training else False       # (one coin per record, in stream order)
                    item = self._cache.get(key)
                    if item is None:
                        item = pending.get(key)          # (a data set shorter than the window: the record is being decoded already)
                        if item is None:
                            item = pending[key] = pool.submit(self._decode, key, rec)
                    window.append((key, item, coin))
                if not window:
                    return
                n = min(self.batch_size, len(window))
                if ring_np is not None:
                    out = ring_np[k % len(ring_np)]
                    slot = ring[k % len(ring)]
                    k += 1
                else:
                    h, w = self.c.image_shape[0], self.c.image_shape[1]
                    out = (np.empty((n, h, w, 3), np.uint8), np.empty((n, self.c.max_num_objects, self.c.num_classes + 1), np.float32),
                           np.empty((n, self.c.max_num_objects, 4), np.float32))
                    slot = tuple(torch.from_numpy(a) for a in out)
                for i in range(n):
                    key, item, coin = window.popleft()
                    if not isinstance(item, tuple):
                        if pending.get(key) is item:
                            del pending[key]
                        item = item.result()
                    image, classes, boxes = self.c._augment(*item, coin)
                    _place_image(out[0][i], image)                                      # (the flip's reversed view is copied here, once)
                    out[1][i], out[2][i] = classes, boxes
                yield tuple(buf[:n] for buf in slot)
                if n < self.batch_size:
                    return
        finally:
            for _, item, _ in window:
                if not isinstance(item, tuple):
                    item.cancel()
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
                              num_workers=None, prefetch=4, shuffle_buffer=1024, cache_bytes=16 << 30):
        """data/input_pipeline.py:19-42.  `filename`: a TFRecord file, a list of them, or a KITTI directory holding
        image_2/ and label_2/.  rank / world_size select this process's shard of the records.
        num_workers: decode threads (default: the cores this process may run on, at most 16); cache_bytes: host memory for decoded
        records that later epochs (and later iterators of the same pipeline: the validation pass of every epoch) reuse; 0 = none."""
        if num_workers is None:
            num_workers = max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)))
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
        return _Dataset(self, sources, batch_size, training, rank, world_size, seed, device, num_workers, prefetch, shuffle_buffer, cache_bytes)

    def _augment(self, image, classes, boxes, do_flip):
        """:43-72 with the coin made by the caller (`do_flip = uniform() > 0.5`)."""
        if not do_flip:
            return image, classes, boxes
        flipped = boxes.copy()
        flipped[:, 0] = np.float32(1.0) - boxes[:, 2]
        flipped[:, 2] = np.float32(1.0) - boxes[:, 0]
        return image[:, ::-1], classes, flipped          # (a reversed VIEW: the batch assembly makes the one copy)

    def _decode_and_preprocess(self, value, do_flip=False):
        """:83-130 for one record: decode, resize, encode the labels, flip."""
        image, classes, boxes = self._augment(*self._decode(value), do_flip)
        return np.ascontiguousarray(image), classes, boxes

    def _decode(self, value):
        """:83-130 up to the augmentation.  `value`: a serialized tf.train.Example, or an (image file, label file) pair of a raw KITTI
        directory.  Returns (image uint8 [H,W,3], classes f32 [100,C+1], boxes f32 [100,4]) -- what the pipeline's cache keeps."""
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
        image = resize_to_uint8(image, new_height, new_width)                            # tf.cast truncates

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
        return image, classes, boxes
