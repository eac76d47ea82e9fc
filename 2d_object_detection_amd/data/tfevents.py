"""TensorBoard scalar event files without TensorFlow.

The reference logs its training / validation scalars with tf.summary.create_file_writer(...) + tf.summary.scalar(tag, value,
step) (train_faster_rcnn.py:102-106,146-154,217-237).  An event file is a TFRecord file (data/tfrecord.py: the same framing and
masked CRC-32C) of `Event` protos; a scalar needs four fields of three messages:

  Event   = { 1: double wall_time, 2: int64 step, 3: string file_version | 5: Summary summary }
  Summary = { 1: repeated Value value }
  Value   = { 1: string tag, 2: float simple_value }

The first record of a file is Event{wall_time, file_version: "brain.Event:2"}.  TensorBoard (and tf.compat.v1.train.summary_iterator)
read these files as they read TensorFlow's own (simple_value is the TF1 scalar form every TensorBoard version plots).
File name: events.out.tfevents.<unix time>.<hostname>.<pid>.<n> as TensorFlow names them."""
import os
import socket
import struct
import time

from .tfrecord import _enc_varint, _fields, _ld, masked_crc32c, read_records


def _event(wall_time, step=None, file_version=None, tag=None, value=None):
    msg = _enc_varint((1 << 3) | 1) + struct.pack("<d", float(wall_time))
    if step is not None:
        msg += _enc_varint((2 << 3) | 0) + _enc_varint(int(step))
    if file_version is not None:
        msg += _ld(3, file_version.encode())
    if tag is not None:
        val = _ld(1, tag.encode()) + _enc_varint((2 << 3) | 5) + struct.pack("<f", float(value))
        msg += _ld(5, _ld(1, val))
    return msg


class EventFileWriter:
    """Append-only scalar event file in `directory` (created).  scalar(tag, value, step) writes and flushes one record."""

    _serial = 0

    def __init__(self, directory):
        os.makedirs(directory, exist_ok=True)
        EventFileWriter._serial += 1
        now = time.time()
        self.path = os.path.join(directory, "events.out.tfevents.%010d.%s.%d.%d" % (int(now), socket.gethostname(), os.getpid(),
                                                                                  EventFileWriter._serial))
        self.fh = open(self.path, "ab")
        self._write(_event(now, file_version="brain.Event:2"))

    def _write(self, data):
        head = struct.pack("<Q", len(data))
        self.fh.write(head + struct.pack("<I", masked_crc32c(head)) + data + struct.pack("<I", masked_crc32c(data)))
        self.fh.flush()

    def scalar(self, tag, value, step):
        self._write(_event(time.time(), step=step, tag=tag, value=value))

    def image(self, tag, png_bytes, height, width, step):
        """tf.summary.image(tag, one RGB image) as TensorBoard stores it: Summary.Value{1: tag, 4: Image{1: height, 2: width, 3: colorspace
        (3 = RGB), 4: encoded_image_string}} (reference train_faster_rcnn.py:180,194)."""
        img = (_enc_varint((1 << 3) | 0) + _enc_varint(int(height)) + _enc_varint((2 << 3) | 0) + _enc_varint(int(width)) +
               _enc_varint((3 << 3) | 0) + _enc_varint(3) + _ld(4, bytes(png_bytes)))
        val = _ld(1, tag.encode()) + _ld(4, img)
        msg = _enc_varint((1 << 3) | 1) + struct.pack("<d", float(time.time())) + _enc_varint((2 << 3) | 0) + _enc_varint(int(step)) + _ld(5, _ld(1, val))
        self._write(msg)

    def close(self):
        if self.fh:
            self.fh.close()
            self.fh = None


def read_scalars(path, verify=True):
    """[(step, tag, value)] of an event file (this writer's or TensorFlow's simple_value scalars); checks the record CRCs."""
    out = []
    for rec in read_records(path, verify=verify):
        step, summary = 0, None
        for field, wire, v in _fields(rec):
            if field == 2 and wire == 0:
                step = v
            elif field == 5 and wire == 2:
                summary = v
        if summary is None:
            continue
        for field, wire, val in _fields(summary):
            if field != 1 or wire != 2:
                continue
            tag, simple = None, None
            for f2, w2, x in _fields(val):
                if f2 == 1 and w2 == 2:
                    tag = bytes(x).decode()
                elif f2 == 2 and w2 == 5:
                    simple = struct.unpack("<f", bytes(x))[0] if not isinstance(x, float) else x
            if tag is not None and simple is not None:
                out.append((int(step), tag, float(simple)))
    return out


def read_images(path, verify=True):
    """[(step, tag, height, width, png bytes)] of the image summaries of an event file."""
    out = []
    for rec in read_records(path, verify=verify):
        step, summary = 0, None
        for field, wire, v in _fields(rec):
            if field == 2 and wire == 0:
                step = v
            elif field == 5 and wire == 2:
                summary = v
        if summary is None:
            continue
        for field, wire, val in _fields(summary):
            if field != 1 or wire != 2:
                continue
            tag, img = None, None
            for f2, w2, x in _fields(val):
                if f2 == 1 and w2 == 2:
                    tag = bytes(x).decode()
                elif f2 == 4 and w2 == 2:
                    img = x
            if tag is None or img is None:
                continue
            h = w = 0
            png = b""
            for f3, w3, y in _fields(img):
                if f3 == 1 and w3 == 0:
                    h = y
                elif f3 == 2 and w3 == 0:
                    w = y
                elif f3 == 4 and w3 == 2:
                    png = bytes(y)
            out.append((int(step), tag, int(h), int(w), png))
    return out
