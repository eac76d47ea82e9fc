"""TFRecord files and tf.train.Example messages without TensorFlow.

The reference stores KITTI as TFRecord files of tf.train.Example protos (data/build_tf_records.py:70-105,128-133) and reads
them back with tf.data.TFRecordDataset + tf.io.parse_single_example (data/input_pipeline.py:31,95-107).  Both formats are
public and tiny, so the files the reference's tooling wrote can be consumed directly:

  record  = uint64 length | uint32 masked_crc32c(length) | bytes data | uint32 masked_crc32c(data)      (little endian)
  Example = { 1: Features { 1: repeated MapEntry { 1: string key, 2: Feature } } }
  Feature = oneof { 1: BytesList { 1: repeated bytes }, 2: FloatList { 1: packed float }, 3: Int64List { 1: packed varint } }

The CRC uses the library's host helper frcnn_crc32c (slicing-by-8); verification on read is optional."""
import struct

import numpy as np

_MASK_DELTA = 0xA282EAD8


def _crc32c(data):
    from .. import _lib
    return int(_lib.load().frcnn_crc32c(0, bytes(data), len(data)))


def masked_crc32c(data):
    crc = _crc32c(data)
    return ((((crc >> 15) | (crc << 17)) & 0xFFFFFFFF) + _MASK_DELTA) & 0xFFFFFFFF


def write_records(path, payloads):
    """Write an iterable of serialized messages as one TFRecord file; returns the number of records."""
    n = 0
    with open(path, "wb") as fh:
        for data in payloads:
            head = struct.pack("<Q", len(data))
            fh.write(head + struct.pack("<I", masked_crc32c(head)) + data + struct.pack("<I", masked_crc32c(data)))
            n += 1
    return n


def read_records(path, verify=False, start=0, step=1):
    """Yield the payloads of records start, start+step, ... (the stride is how data-parallel ranks shard one file);
    skipped records are seeked over, not read."""
    with open(path, "rb") as fh:
        i = 0
        while True:
            head = fh.read(12)
            if not head:
                return
            if len(head) < 12:
                raise ValueError("%s: truncated record header" % path)
            (length,) = struct.unpack("<Q", head[:8])
            if verify and struct.unpack("<I", head[8:])[0] != masked_crc32c(head[:8]):
                raise ValueError("%s: corrupt length field in record %d" % (path, i))
            mine = i >= start and (i - start) % step == 0
            if mine:
                data = fh.read(length)
                tail = fh.read(4)
                if len(data) < length or len(tail) < 4:
                    raise ValueError("%s: truncated record %d" % (path, i))
                if verify and struct.unpack("<I", tail)[0] != masked_crc32c(data):
                    raise ValueError("%s: corrupt data in record %d" % (path, i))
                yield data
            else:
                fh.seek(length + 4, 1)
            i += 1


# ---------------------------------------------------------------------------------------------------- protobuf wire format
def _varint(buf, pos):
    result = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _fields(buf):
    """Yield (field number, wire type, value) of one message; length-delimited values come as memoryviews."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v, pos = buf[pos:pos + 8], pos + 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v, pos = buf[pos:pos + ln], pos + ln
        elif wt == 5:
            v, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield field, wt, v


def _signed64(v):
    return v - (1 << 64) if v >= (1 << 63) else v


def _parse_feature(buf):
    for kind, _, body in _fields(buf):
        if kind == 1:                                   # BytesList
            return [bytes(v) for f, _, v in _fields(body) if f == 1]
        if kind == 2:                                   # FloatList: packed (wire type 2) or one fixed32 per value
            out = []
            for f, wt, v in _fields(body):
                if f == 1:
                    out.append(np.frombuffer(bytes(v), dtype="<f4"))
            return np.concatenate(out).astype(np.float32) if out else np.zeros(0, np.float32)
        if kind == 3:                                   # Int64List: packed varints or one varint per value
            out = []
            for f, wt, v in _fields(body):
                if f != 1:
                    continue
                if wt == 0:
                    out.append(_signed64(v))
                else:
                    pos = 0
                    while pos < len(v):
                        x, pos = _varint(v, pos)
                        out.append(_signed64(x))
            return np.asarray(out, dtype=np.int64)
    return []


def parse_example(data):
    """Serialized tf.train.Example -> {feature name: list of bytes | float32 array | int64 array}."""
    out = {}
    buf = memoryview(data)
    for f, _, features in _fields(buf):
        if f != 1:
            continue
        for g, _, entry in _fields(features):
            if g != 1:
                continue
            key, value = None, []
            for h, _, v in _fields(entry):
                if h == 1:
                    key = bytes(v).decode()
                elif h == 2:
                    value = _parse_feature(v)
            out[key] = value
    return out


def _enc_varint(v):
    v &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(field, payload):
    return _enc_varint((field << 3) | 2) + _enc_varint(len(payload)) + payload


def serialize_example(features):
    """{name: bytes | list of bytes | float array/list | int array/list} -> serialized tf.train.Example (packed lists)."""
    entries = b""
    for key in sorted(features):
        v = features[key]
        if isinstance(v, (bytes, bytearray)):
            v = [bytes(v)]
        if len(v) and isinstance(v[0], (bytes, bytearray)):
            feat = _ld(1, b"".join(_ld(1, bytes(x)) for x in v))
        else:
            arr = np.asarray(v)
            if arr.dtype.kind == "f":
                feat = _ld(2, _ld(1, arr.astype("<f4").tobytes()) if arr.size else b"")
            else:
                feat = _ld(3, _ld(1, b"".join(_enc_varint(int(x)) for x in arr.reshape(-1))) if arr.size else b"")
        entries += _ld(1, _ld(1, key.encode()) + _ld(2, feat))
    return _ld(1, entries)
