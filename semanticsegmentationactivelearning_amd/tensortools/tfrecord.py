"""TFRecord / tf.train.Example access without TensorFlow: mirror of the reference's
``tensortools/tfrecord.py`` (record framing :9-18, iterator :40-52, single-record reader :67-76) plus
the minimal protobuf wire codec the default parser of ``tensortools/input.py:165-172`` needs.

Record framing (reference tfrecord.py:9-18):
    uint64 length | uint32 masked-crc32c(length) | bytes[length] | uint32 masked-crc32c(data)
The reference's pure-python readers skip both CRCs (:19-20); so does this reader unless
``check_crc=True``.  The writer below (used by the tests and by anyone packing synthetic pools)
emits valid CRCs so the files are readable by real TensorFlow.

tf.train.Example wire format (proto3):
    Example  { Features features = 1; }
    Features { map<string, Feature> feature = 1; }          # map entry: key = 1, value = 2
    Feature  { oneof kind { BytesList bytes_list = 1; FloatList float_list = 2; Int64List int64_list = 3; } }
    BytesList { repeated bytes value = 1; }   FloatList { repeated float value = 1 [packed]; }
    Int64List { repeated int64 value = 1 [packed]; }
Dataset schema (reference README.md:18-42, writer generate_dataset.py:188-221): ``image/data``,
``image/encoding``, ``image/channels``, ``label``, ``height``, ``width``, ``id`` and optional
``<modality>/{data,encoding,channels}``.
"""
import struct

import numpy as np

# ---- CRC32C (Castagnoli), table driven, + TFRecord masking ----------------------------------------
_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        tbl = np.zeros(256, dtype=np.uint32)
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            tbl[i] = c
        _CRC_TABLE = tbl.tolist()
    return _CRC_TABLE


def crc32c(data):
    tbl = _crc_table()
    c = 0xFFFFFFFF
    for b in data:
        c = tbl[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data):
    c = crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


# ---- record framing ---------------------------------------------------------------------------------
def tfrecord_iterator(filename, check_crc=False):
    """yields the raw bytes of every record (reference tfrecord.py:40-52)"""
    with open(filename, "rb") as f:
        while True:
            header = f.read(12)
            if header == b"":
                break
            if len(header) < 12:
                raise ValueError("%s: truncated record header" % filename)
            record_length = struct.unpack("<Q", header[:8])[0]
            data = f.read(record_length)
            footer = f.read(4)
            if len(data) < record_length or len(footer) < 4:
                raise ValueError("%s: truncated record" % filename)
            if check_crc:
                if struct.unpack("<I", header[8:])[0] != masked_crc32c(header[:8]):
                    raise ValueError("%s: length CRC mismatch" % filename)
                if struct.unpack("<I", footer)[0] != masked_crc32c(data):
                    raise ValueError("%s: data CRC mismatch" % filename)
            yield data


def read_tfrecord(filename):
    """first serialized record of a file, b"" if the file is empty (reference tfrecord.py:67-76)"""
    for rec in tfrecord_iterator(filename):
        return rec
    return b""


def write_tfrecord(filename, records):
    """TFRecord writer with valid masked CRC32C (what tf.io.TFRecordWriter emits)"""
    with open(filename, "wb") as f:
        for rec in records:
            length = struct.pack("<Q", len(rec))
            f.write(length)
            f.write(struct.pack("<I", masked_crc32c(length)))
            f.write(rec)
            f.write(struct.pack("<I", masked_crc32c(rec)))


# ---- protobuf wire codec for tf.train.Example ----------------------------------------------------------
def _read_varint(buf, pos):
    result, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 70:
            raise ValueError("malformed varint")


def _fields(buf):
    """yields (field_number, wire_type, value) of one message; value is int or memoryview"""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _read_varint(buf, pos)
        field, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _read_varint(buf, pos)
        elif wt == 1:
            val, pos = buf[pos:pos + 8], pos + 8
        elif wt == 2:
            ln, pos = _read_varint(buf, pos)
            val, pos = buf[pos:pos + ln], pos + ln
        elif wt == 5:
            val, pos = buf[pos:pos + 4], pos + 4
        else:
            raise ValueError("unsupported wire type %d" % wt)
        yield field, wt, val


def _parse_feature(buf):
    for field, wt, val in _fields(buf):
        if field == 1:  # BytesList
            return "bytes_list", [bytes(v) for f, w, v in _fields(val) if f == 1]
        if field == 2:  # FloatList (packed or not)
            out = []
            for f, w, v in _fields(val):
                if f == 1 and w == 2:
                    out.extend(struct.unpack("<%df" % (len(v) // 4), bytes(v)))
                elif f == 1 and w == 5:
                    out.append(struct.unpack("<f", bytes(v))[0])
            return "float_list", out
        if field == 3:  # Int64List (packed or not)
            out = []
            for f, w, v in _fields(val):
                if f == 1 and w == 2:
                    p = 0
                    while p < len(v):
                        x, p = _read_varint(v, p)
                        out.append(x - (1 << 64) if x >> 63 else x)
                elif f == 1 and w == 0:
                    out.append(v - (1 << 64) if v >> 63 else v)
            return "int64_list", out
    return None, []


def parse_example(serialized):
    """serialized tf.train.Example -> {key: (kind, [values])}"""
    buf = memoryview(serialized)
    out = {}
    for field, wt, features in _fields(buf):
        if field != 1:
            continue
        for f2, w2, entry in _fields(features):
            if f2 != 1:
                continue
            key, feat = None, None
            for f3, w3, v in _fields(entry):
                if f3 == 1:
                    key = bytes(v).decode("utf-8")
                elif f3 == 2:
                    feat = _parse_feature(v)
            if key is not None and feat is not None:
                out[key] = feat
    return out


def parse_single_example(serialized, fmt):
    """tf.io.parse_single_example with FixedLenFeature(()) entries: ``fmt`` maps key -> default
    (bytes default for string features, int for int64); missing keys take the default
    (reference input.py:165-172)."""
    ex = parse_example(serialized)
    out = {}
    for key, default in fmt.items():
        kind, vals = ex.get(key, (None, []))
        out[key] = vals[0] if vals else default
    return out


def tfrecord2example_dict(filename):
    """first Example of a file as a plain dict (reference tfrecord.py:78-79, MessageToDict-like)"""
    ex = parse_example(read_tfrecord(filename))
    return {"features": {"feature": {k: {{"bytes_list": "bytesList", "float_list": "floatList",
                                          "int64_list": "int64List"}[kind]: {"value": vals}}
                                     for k, (kind, vals) in ex.items()}}}


def _varint(x):
    if x < 0:
        x += 1 << 64
    out = bytearray()
    while True:
        b = x & 0x7F
        x >>= 7
        if x:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _ld(field, payload):
    return _varint((field << 3) | 2) + _varint(len(payload)) + payload


def make_example(features):
    """{key: bytes | str | int | float | list thereof} -> serialized tf.train.Example"""
    entries = b""
    for key in sorted(features):
        val = features[key]
        vals = val if isinstance(val, (list, tuple)) else [val]
        if all(isinstance(v, (bytes, bytearray, str)) for v in vals):
            payload = b"".join(_ld(1, v.encode() if isinstance(v, str) else bytes(v)) for v in vals)
            feat = _ld(1, payload)
        elif all(isinstance(v, (int, np.integer)) for v in vals):
            feat = _ld(3, _ld(1, b"".join(_varint(int(v)) for v in vals)))
        else:
            feat = _ld(2, _ld(1, struct.pack("<%df" % len(vals), *[float(v) for v in vals])))
        entries += _ld(1, _ld(1, key.encode()) + _ld(2, feat))
    return _ld(1, entries)
