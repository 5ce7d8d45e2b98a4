"""TEST INFRASTRUCTURE: a second, independent encoder of TensorFlow's V2 checkpoint format ("tensor bundle"), written from the format
definitions alone — tensorflow/core/protobuf/tensor_bundle.proto (BundleHeaderProto / BundleEntryProto), tensor_shape.proto, and LevelDB's
doc/table_format.md for the `.index` file — and deliberately different in every free choice from dan_amd/utility/checkpoint.py's own
writer, which it shares no code with: TWO data shards, one table entry per restart point (restart interval 1, so no key prefix is ever
shared), one data block per ~700 bytes of entries, separator keys = the block's last key, a CRC over every tensor.  What the repo's READER
accepts from this file it would accept from TensorFlow's writer for the same reasons (the released checkpoints are
TF-written; none exists offline, so this stands in for "a file the repo did not write itself")."""
import struct

import numpy as np

_MAGIC = 0xDB4775248B80FB57
_DT = {np.dtype("float32"): 1, np.dtype("int32"): 3, np.dtype("int64"): 9}


def _crc32c_tables():
    poly = 0x82F63B78                     # Castagnoli, reflected
    t = []
    for i in range(256):
        c = i
        for _ in range(8):
            c = (c >> 1) ^ (poly if (c & 1) else 0)
        t.append(c)
    return t


_T = _crc32c_tables()


def crc32c(data):
    c = 0xFFFFFFFF
    t = _T
    for b in data:
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked(c):
    return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF


def varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _field_varint(no, v):
    return varint(no << 3) + varint(v)


def _field_bytes(no, b):
    return varint((no << 3) | 2) + varint(len(b)) + b


def _field_fixed32(no, v):
    return varint((no << 3) | 5) + struct.pack("<I", v)


def _shape_proto(shape):
    return b"".join(_field_bytes(2, _field_varint(1, int(d))) for d in shape)


def _block(entries):
    """LevelDB block with restart interval 1: every entry stores its whole key (shared = 0) and is a restart point."""
    body, restarts = bytearray(), []
    for k, v in entries:
        restarts.append(len(body))
        body += varint(0) + varint(len(k)) + varint(len(v)) + k + v
    for r in restarts:
        body += struct.pack("<I", r)
    body += struct.pack("<I", len(restarts))
    return bytes(body)


def _emit(out, block):
    """Appends block + trailer (type 0 = uncompressed, masked crc32c over block + type byte); -> BlockHandle bytes."""
    handle = varint(len(out)) + varint(len(block))
    out += block + b"\x00" + struct.pack("<I", masked(crc32c(block + b"\x00")))
    return handle


def write_bundle(prefix, tensors, shards=2, entries_per_block_bytes=700):
    names = sorted(tensors, key=lambda s: s.encode("utf-8"))
    data = [bytearray() for _ in range(shards)]
    entries = [(b"", _field_varint(1, shards) + _field_bytes(3, _field_varint(1, 1)))]       # header: num_shards, little endian (default), version.producer
    for i, n in enumerate(names):
        a = np.asarray(tensors[n])                      # (ascontiguousarray would turn a scalar into shape (1,))
        raw = a.astype(a.dtype.newbyteorder("<")).tobytes()
        sh = i % shards
        e = _field_varint(1, _DT[a.dtype]) + _field_bytes(2, _shape_proto(a.shape))
        if sh:
            e += _field_varint(3, sh)
        if len(data[sh]):
            e += _field_varint(4, len(data[sh]))
        e += _field_varint(5, len(raw)) + _field_fixed32(6, masked(crc32c(raw)))
        data[sh] += raw
        entries.append((n.encode("utf-8"), e))
    out, index, cur, size = bytearray(), [], [], 0
    for k, v in entries:
        cur.append((k, v))
        size += len(k) + len(v)
        if size >= entries_per_block_bytes:
            index.append((cur[-1][0], _emit(out, _block(cur))))
            cur, size = [], 0
    if cur:
        index.append((cur[-1][0], _emit(out, _block(cur))))
    meta = _emit(out, _block([]))
    idx = _emit(out, _block(index))
    footer = meta + idx
    out += footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", _MAGIC)
    with open(prefix + ".index", "wb") as f:
        f.write(out)
    for s in range(shards):
        with open("%s.data-%05d-of-%05d" % (prefix, s, shards), "wb") as f:
            f.write(data[s])
