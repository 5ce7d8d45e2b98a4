"""WIDER-FACE TFRecord input — mirrors dataset/dataset_common.py (slim_get_batch :33-193) over the record schema written by
dataset/convert_tfrecords.py:122-139, without TensorFlow: the TFRecord framing, the tf.Example protobuf wire format and the slim
decoders (Image / Tensor / BoundingBox handlers, :77-91) are restated here on bytes; the queues of DatasetDataProvider /
tf.train.maybe_shuffle_batch (:103-116, :178-193) become one bounded shuffle buffer on the host.

    record   = u64 length | u32 masked_crc32c(length) | bytes[length] | u32 masked_crc32c(bytes)         (little endian)
    Example  = { 1: Features{ 1: map<string, Feature> } };  Feature = oneof { 1: BytesList, 2: FloatList, 3: Int64List }, lists in field 1
               (floats / int64s packed or unpacked)

Host I/O only: images are decoded with Pillow (the one JPEG decoder in this image); the batch leaves as pinned uint8 / fp32 tensors for
the device-side augmentation and anchor encoding (preprocessing/dan_preprocessing.py, utility/anchor_manipulator.py)."""
import glob
import io
import random
import struct

import numpy as np

# use dataset_inspect.py to get these summary (dataset_common.py:27-31)
data_splits_num = {
    'train': 12880,
    'valid': 3226,
    '?????': 12880 + 3226,
}

# ---------------------------------------------------------------------------------------------------------- TFRecord framing
_CRC_TABLE = None


def _crc32c(data):
    """CRC-32C (Castagnoli), the checksum of the TFRecord framing."""
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            t.append(c)
        _CRC_TABLE = t
    c = 0xFFFFFFFF
    t = _CRC_TABLE
    for b in data:
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def masked_crc32c(data):
    c = _crc32c(data)
    return ((((c >> 15) | (c << 17)) & 0xFFFFFFFF) + 0xA282EAD8) & 0xFFFFFFFF


def read_tfrecord(path, verify_payload=False):
    """Yields the payload of every record.  The length checksum is always verified (a wrong length would desynchronise the stream);
    the payload checksum only on request (pure-Python CRC over a JPEG costs more than decoding it)."""
    with open(path, "rb") as f:
        while True:
            head = f.read(12)
            if not head:
                return
            if len(head) < 12:
                raise IOError("%s: truncated record header" % path)
            n, = struct.unpack("<Q", head[:8])
            if struct.unpack("<I", head[8:])[0] != masked_crc32c(head[:8]):
                raise IOError("%s: corrupt record length" % path)
            payload = f.read(n)
            tail = f.read(4)
            if len(payload) < n or len(tail) < 4:
                raise IOError("%s: truncated record" % path)
            if verify_payload and struct.unpack("<I", tail)[0] != masked_crc32c(payload):
                raise IOError("%s: corrupt record payload" % path)
            yield payload


def write_tfrecord(path, payloads):
    with open(path, "wb") as f:
        for p in payloads:
            head = struct.pack("<Q", len(p))
            f.write(head + struct.pack("<I", masked_crc32c(head)) + p + struct.pack("<I", masked_crc32c(p)))


# ---------------------------------------------------------------------------------------------------------- tf.Example wire format
def _varint(buf, pos):
    v, shift = 0, 0
    while True:
        b = buf[pos]
        pos += 1
        v |= (b & 0x7F) << shift
        if not b & 0x80:
            return v, pos
        shift += 7


def _fields(buf):
    """(field number, wire type, value) of one message; value = int (varint / fixed) or memoryview (length delimited)."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield fno, wt, v


def _feature(buf):
    for fno, wt, v in _fields(buf):
        if wt != 2:
            continue
        if fno == 1:                                                     # BytesList
            return [bytes(x) for f, w, x in _fields(v) if f == 1 and w == 2]
        if fno == 2:                                                     # FloatList: packed (one blob) or one fixed32 per value
            out = []
            for f, w, x in _fields(v):
                if f != 1:
                    continue
                if w == 2:
                    out.append(np.frombuffer(bytes(x), dtype="<f4"))
                else:
                    out.append(np.frombuffer(struct.pack("<I", x), dtype="<f4"))
            return np.concatenate(out).astype(np.float32) if out else np.zeros((0,), np.float32)
        if fno == 3:                                                     # Int64List: packed varints or one varint per value
            out = []
            for f, w, x in _fields(v):
                if f != 1:
                    continue
                if w == 2:
                    p = 0
                    while p < len(x):
                        val, p = _varint(x, p)
                        out.append(val)
                else:
                    out.append(x)
            return np.asarray([o - (1 << 64) if o >= (1 << 63) else o for o in out], dtype=np.int64)
    return []


def parse_example(payload):
    """tf.Example bytes -> {feature name: list of bytes | float32 array | int64 array}."""
    buf = memoryview(payload)
    feats = {}
    for fno, wt, v in _fields(buf):
        if fno != 1 or wt != 2:
            continue
        for f2, w2, entry in _fields(v):                                 # Features.feature map entries
            if f2 != 1 or w2 != 2:
                continue
            key, val = None, []
            for f3, w3, x in _fields(entry):
                if f3 == 1 and w3 == 2:
                    key = bytes(x).decode("utf8")
                elif f3 == 2 and w3 == 2:
                    val = _feature(x)
            if key is not None:
                feats[key] = val
    return feats


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


def _ld(fno, data):
    return _enc_varint((fno << 3) | 2) + _enc_varint(len(data)) + data


def convert_to_example(image_name, image_buffer, bboxes, blur, expression, illumination, invalid, occlusion, pose, height, width):
    """dataset/convert_tfrecords.py:90-139: the serialized tf.Example of one image (bboxes: [ymin, xmin, ymax, xmax] rows, normalised)."""
    ymin, xmin, ymax, xmax = ([float(b[i]) for b in bboxes] for i in range(4))

    def ints(v):
        v = list(v) if isinstance(v, (list, tuple, np.ndarray)) else [v]
        return _ld(3, _ld(1, b"".join(_enc_varint(int(x)) for x in v)))

    def floats(v):
        return _ld(2, _ld(1, np.asarray(v, dtype="<f4").tobytes()))

    def byts(v):
        return _ld(1, _ld(1, v))

    feature = {
        'image/height': ints(height), 'image/width': ints(width), 'image/channels': ints(3), 'image/shape': ints([height, width, 3]),
        'image/object/bbox/xmin': floats(xmin), 'image/object/bbox/xmax': floats(xmax), 'image/object/bbox/ymin': floats(ymin),
        'image/object/bbox/ymax': floats(ymax), 'image/object/bbox/blur': ints(blur), 'image/object/bbox/expression': ints(expression),
        'image/object/bbox/illumination': ints(illumination), 'image/object/bbox/invalid': ints(invalid),
        'image/object/bbox/occlusion': ints(occlusion), 'image/object/bbox/pose': ints(pose), 'image/format': byts(b"JPEG"),
        'image/filename': byts(image_name.encode("utf8")), 'image/encoded': byts(image_buffer),
    }
    entries = b"".join(_ld(1, _ld(1, k.encode("utf8")) + _ld(2, v)) for k, v in feature.items())
    return _ld(1, entries)


# ---------------------------------------------------------------------------------------------------------- slim decoders (:77-91)
def decode_image(encoded, fmt=b"jpeg"):
    """slim.tfexample_decoder.Image('image/encoded', 'image/format'): uint8 [H,W,3] RGB."""
    try:
        from PIL import Image
    except ImportError as e:                                             # fail loudly: there is no second decoder to fall back to
        raise RuntimeError("decoding %r records needs Pillow" % fmt) from e
    with Image.open(io.BytesIO(encoded)) as im:
        return np.asarray(im.convert("RGB"), dtype=np.uint8)


def decode_record(payload, decode=True):
    """One record -> the items of items_to_handlers (:77-91); 'object/bbox' rows are [ymin, xmin, ymax, xmax] (BoundingBox handler)."""
    f = parse_example(payload)

    def one(key, default=b""):
        v = f.get(key, [])
        return v[0] if len(v) else default

    def arr(key, dtype):
        v = f.get(key, [])
        return np.asarray(v, dtype=dtype) if len(v) else np.zeros((0,), dtype)

    box = [arr('image/object/bbox/' + k, np.float32) for k in ('ymin', 'xmin', 'ymax', 'xmax')]
    item = {
        'filename': one('image/filename'),
        'shape': arr('image/shape', np.int64),
        'object/bbox': np.stack(box, axis=1) if len(box[0]) else np.zeros((0, 4), np.float32),
    }
    for k in ('blur', 'expression', 'illumination', 'invalid', 'occlusion', 'pose'):
        item['object/' + k] = arr('image/object/bbox/' + k, np.int64)
    item['image'] = decode_image(one('image/encoded'), one('image/format', b"jpeg")) if decode else one('image/encoded')
    return item


# ---------------------------------------------------------------------------------------------------------- the batch generator
def slim_get_batch(num_classes, batch_size, split_name, file_pattern, num_readers, num_preprocessing_threads, image_preprocessing_fn,
                   anchor_encoder, num_epochs=None, is_training=True, seed=None):
    """dataset_common.py:33-193 as a generator of batches (lists of per-image entries; stacking is the caller's, the anchor encoder's
    outputs have fixed shapes).

    Training entries: [image, filename, shape] + gt_targets + gt_labels + gt_scores + gt_bboxes (:152-170) with
    image, gbboxes = image_preprocessing_fn(org_image, g_bboxes) and the four groups = anchor_encoder(gbboxes); records are drawn from a
    shuffle buffer of capacity 64 * batch_size once it holds 8 * batch_size (:178-186) and an image whose boxes are all gone after
    the augmentation is skipped (keep_input, :182).  Evaluation entries: [image, filename, shape, output_shape, gbboxes] (:173-176) in
    file order, final batch allowed to be smaller (:188-193).
    num_readers / num_preprocessing_threads are accepted for signature parity; reading is sequential here."""
    if split_name not in data_splits_num:
        raise ValueError('split name %s was not recognized.' % split_name)
    files = sorted(glob.glob(file_pattern.format(split_name)))
    if not files:
        raise IOError("no record file matches %r" % file_pattern.format(split_name))
    rng = random.Random(seed)

    def records():
        epoch = 0
        while num_epochs is None or epoch < num_epochs:
            order = list(files)
            if is_training:
                rng.shuffle(order)
            for path in order:
                for payload in read_tfrecord(path):
                    yield payload
            epoch += 1

    def entry(payload):
        item = decode_record(payload)
        g_bboxes = item['object/bbox']
        if is_training:
            # isinvalid_mask = tf.ones_like(g_invalid < 1) (:123): every annotated face is kept
            image, gbboxes = image_preprocessing_fn(item['image'], g_bboxes)
            if len(gbboxes) == 0:
                return None                                              # keep_input = (tf.shape(gbboxes)[0] > 0)
            groups = anchor_encoder(gbboxes)
            out = [image, item['filename'], item['shape']]
            for grp in groups:
                out += list(grp) if isinstance(grp, (list, tuple)) else [grp]
            return out
        image, output_shape = image_preprocessing_fn(item['image'], g_bboxes)
        if len(g_bboxes) == 0:
            return None
        return [image, item['filename'], item['shape'], output_shape, g_bboxes]

    capacity, min_after = 64 * batch_size, 8 * batch_size
    buf, batch = [], []
    for payload in records():
        if is_training:
            buf.append(payload)
            if len(buf) < min(capacity, min_after + batch_size):
                continue
            payload = buf.pop(rng.randrange(len(buf)))
        e = entry(payload)
        if e is None:
            continue
        batch.append(e)
        if len(batch) == batch_size:
            yield batch
            batch = []
    while buf:                                                           # drain the shuffle buffer at the end of the last epoch
        e = entry(buf.pop(rng.randrange(len(buf))))
        if e is None:
            continue
        batch.append(e)
        if len(batch) == batch_size:
            yield batch
            batch = []
    if batch and not is_training:                                        # allow_smaller_final_batch = (not is_training)
        yield batch
