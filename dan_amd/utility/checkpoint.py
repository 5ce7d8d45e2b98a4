"""TensorFlow V2 checkpoint ("tensor bundle") reader / writer and the reference's warm-start rules.

The reference restores `vgg16_reducedfc.ckpt` / its own checkpoints through tf.train.Saver (utility/scaffolds.py:24-88, called from
train_sfd.py:159-163, train_dan.py, train_pb.py with name_remap {'/conv2d/kernel': '/weights', '/conv2d/bias': '/biases'}).
TensorFlow is not available here, so the on-disk format is read directly:

  <prefix>.index                 an SSTable (TF's port of the LevelDB table format): 48-byte footer with the metaindex and
                                 index block handles + magic 0xdb4775248b80fb57; blocks = prefix-compressed entries + restart
                                 array, followed by a 1-byte compression tag (0 none, 1 snappy) and a masked crc32c.
                                 key ""      -> BundleHeaderProto  {1: num_shards, 2: endianness, 3: version}
                                 key <name>  -> BundleEntryProto   {1: dtype, 2: shape{2: dim{1: size}}, 3: shard_id, 4: offset,
                                                                    5: size, 6: crc32c (fixed32), 7: slices}
  <prefix>.data-0000i-of-0000N   raw little-endian tensor bytes.

`CheckpointReader` mirrors tf.train.NewCheckpointReader (has_tensor / get_tensor / get_variable_to_shape_map), `latest_checkpoint`
mirrors tf.train.latest_checkpoint, and `init_from_checkpoint` applies scaffolds.get_init_fn_for_scaffold's scope / exclusion /
remap / missing-variable rules to a VariableStore.  Kernel layouts need no conversion: conv kernels are HWIO in both, the
deformable-conv kernel is OIHW in both (custom_op.py:134).  Host-side code (numpy); nothing here touches the GPU path.
"""
import os
import struct

import numpy as np

TABLE_MAGIC = 0xdb4775248b80fb57
# tensorflow/core/framework/types.proto
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_, 17: np.uint16,
          19: np.float16, 22: np.uint32, 23: np.uint64}
DTYPE_ENUM = {np.dtype(v): k for k, v in DTYPES.items()}


class CheckpointError(ValueError):
    pass


# ------------------------------------------------------------------------------------------------------------ crc32c
_CRC_TABLE = None


def _crc_table():
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = []
        for i in range(256):
            c = i
            for _ in range(8):
                c = (c >> 1) ^ 0x82F63B78 if c & 1 else c >> 1
            t.append(c)
        _CRC_TABLE = t
    return _CRC_TABLE


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli), as used by the table blocks and the bundle entries."""
    t = _crc_table()
    c = crc ^ 0xFFFFFFFF
    for b in bytes(data):
        c = t[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def mask_crc(c):
    return (((c >> 15) | (c << 17)) + 0xa282ead8) & 0xFFFFFFFF


# ------------------------------------------------------------------------------------------------------------ varints / protobuf
def _varint(buf, pos):
    shift = result = 0
    while True:
        if pos >= len(buf):
            raise CheckpointError("truncated varint")
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _put_varint(v):
    out = bytearray()
    while True:
        b = v & 0x7F
        v >>= 7
        if v:
            out.append(b | 0x80)
        else:
            out.append(b)
            return bytes(out)


def _proto_fields(buf):
    """Yields (field number, wire type, value) of one protobuf message (value: int for varint/fixed, bytes for length-delimited)."""
    pos = 0
    while pos < len(buf):
        tag, pos = _varint(buf, pos)
        f, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = struct.unpack_from("<Q", buf, pos)[0]
            pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            v = bytes(buf[pos:pos + n])
            pos += n
        elif wt == 5:
            v = struct.unpack_from("<I", buf, pos)[0]
            pos += 4
        else:
            raise CheckpointError("unsupported protobuf wire type %d" % wt)
        yield f, wt, v


def _parse_shape(buf):
    dims = []
    for f, _, v in _proto_fields(buf):
        if f == 2:                                        # Dim
            size = 0
            for g, _, w in _proto_fields(v):
                if g == 1:
                    size = w if w < (1 << 63) else w - (1 << 64)
            dims.append(size)
        elif f == 3 and v:
            raise CheckpointError("tensor of unknown rank in checkpoint")
    return tuple(dims)


# ------------------------------------------------------------------------------------------------------------ snappy (blocks may be compressed)
def snappy_decompress(buf):
    n, pos = _varint(buf, 0)
    out = bytearray()
    while pos < len(buf):
        tag = buf[pos]
        pos += 1
        kind = tag & 3
        if kind == 0:                                     # literal
            ln = tag >> 2
            if ln >= 60:
                nb = ln - 59
                ln = int.from_bytes(buf[pos:pos + nb], "little")
                pos += nb
            ln += 1
            out += buf[pos:pos + ln]
            pos += ln
            continue
        if kind == 1:
            ln = ((tag >> 2) & 7) + 4
            off = ((tag >> 5) << 8) | buf[pos]
            pos += 1
        elif kind == 2:
            ln = (tag >> 2) + 1
            off = buf[pos] | (buf[pos + 1] << 8)
            pos += 2
        else:
            ln = (tag >> 2) + 1
            off = int.from_bytes(buf[pos:pos + 4], "little")
            pos += 4
        if off == 0 or off > len(out):
            raise CheckpointError("corrupt snappy stream")
        for _ in range(ln):                               # copies may overlap their own output
            out.append(out[-off])
    if len(out) != n:
        raise CheckpointError("snappy length mismatch")
    return bytes(out)


# ------------------------------------------------------------------------------------------------------------ table (index file)
def _read_block(data, offset, size, verify=True):
    raw = data[offset:offset + size]
    trailer = data[offset + size:offset + size + 5]
    if len(raw) != size or len(trailer) != 5:
        raise CheckpointError("block handle beyond the end of the index file")
    if verify:
        want = struct.unpack("<I", trailer[1:])[0]
        if mask_crc(crc32c(raw + trailer[:1])) != want:
            raise CheckpointError("index block checksum mismatch")
    if trailer[0] == 0:
        return raw
    if trailer[0] == 1:
        return snappy_decompress(raw)
    raise CheckpointError("unknown block compression %d" % trailer[0])


def _block_entries(block):
    if len(block) < 4:
        raise CheckpointError("short block")
    nrestart = struct.unpack_from("<I", block, len(block) - 4)[0]
    end = len(block) - 4 - 4 * nrestart
    pos, key = 0, b""
    while pos < end:
        shared, pos = _varint(block, pos)
        non_shared, pos = _varint(block, pos)
        vlen, pos = _varint(block, pos)
        key = key[:shared] + bytes(block[pos:pos + non_shared])
        pos += non_shared
        yield key, bytes(block[pos:pos + vlen])
        pos += vlen


def read_table(path, verify=True):
    """-> list of (key bytes, value bytes) of an SSTable file, in key order."""
    data = open(path, "rb").read()
    if len(data) < 48:
        raise CheckpointError("%s: too short for a table footer" % path)
    footer = data[-48:]
    if struct.unpack("<Q", footer[40:])[0] != TABLE_MAGIC:
        raise CheckpointError("%s: bad table magic (not a TF V2 checkpoint index)" % path)
    p = 0
    _, p = _varint(footer, p)                             # metaindex handle (unused)
    _, p = _varint(footer, p)
    ioff, p = _varint(footer, p)
    isize, p = _varint(footer, p)
    out = []
    for _, handle in _block_entries(_read_block(data, ioff, isize, verify)):
        off, q = _varint(handle, 0)
        size, q = _varint(handle, q)
        out.extend(_block_entries(_read_block(data, off, size, verify)))
    return out


# ------------------------------------------------------------------------------------------------------------ reader
class CheckpointReader(object):
    """tf.train.NewCheckpointReader for V2 checkpoints given by their prefix (e.g. './model/vgg16_reducedfc.ckpt')."""

    def __init__(self, prefix, verify_index=True):
        self.prefix = prefix
        index = prefix + ".index"
        if not os.path.exists(index):
            raise CheckpointError("%s not found (V1 single-file checkpoints are not supported)" % index)
        self.num_shards, self.entries = 1, {}
        for key, val in read_table(index, verify_index):
            if key == b"":
                for f, _, v in _proto_fields(val):
                    if f == 1:
                        self.num_shards = v
                    elif f == 2 and v != 0:
                        raise CheckpointError("big-endian checkpoint")
                continue
            e = {"dtype": 0, "shape": (), "shard": 0, "offset": 0, "size": 0, "crc": None, "sliced": False}
            for f, _, v in _proto_fields(val):
                if f == 1:
                    e["dtype"] = v
                elif f == 2:
                    e["shape"] = _parse_shape(v)
                elif f == 3:
                    e["shard"] = v
                elif f == 4:
                    e["offset"] = v
                elif f == 5:
                    e["size"] = v
                elif f == 6:
                    e["crc"] = v
                elif f == 7:
                    e["sliced"] = True
            self.entries[key.decode("utf-8")] = e

    def has_tensor(self, name):
        return name in self.entries

    def get_variable_to_shape_map(self):
        return {n: list(e["shape"]) for n, e in self.entries.items()}

    def get_variable_to_dtype_map(self):
        return {n: DTYPES.get(e["dtype"]) for n, e in self.entries.items()}

    def get_tensor(self, name, verify=False):
        if name not in self.entries:
            raise KeyError("tensor %r not found in checkpoint %s" % (name, self.prefix))
        e = self.entries[name]
        if e["sliced"]:
            raise CheckpointError("%s is a partitioned variable (slices are not supported)" % name)
        if e["dtype"] not in DTYPES:
            raise CheckpointError("%s: unsupported dtype enum %d" % (name, e["dtype"]))
        path = "%s.data-%05d-of-%05d" % (self.prefix, e["shard"], self.num_shards)
        with open(path, "rb") as f:
            f.seek(e["offset"])
            raw = f.read(e["size"])
        dt = np.dtype(DTYPES[e["dtype"]])
        count = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
        if len(raw) != e["size"] or count * dt.itemsize != e["size"]:
            raise CheckpointError("%s: %d bytes on disk for shape %s %s" % (name, len(raw), e["shape"], dt))
        if verify and e["crc"] is not None and mask_crc(crc32c(raw)) != e["crc"]:
            raise CheckpointError("%s: data checksum mismatch" % name)
        return np.frombuffer(raw, dtype=dt.newbyteorder("<")).reshape(e["shape"]).astype(dt)


def latest_checkpoint(checkpoint_dir):
    """tf.train.latest_checkpoint: the prefix named by the 'checkpoint' state file, or None."""
    state = os.path.join(checkpoint_dir, "checkpoint")
    if not os.path.exists(state):
        return None
    for line in open(state):
        line = line.strip()
        if line.startswith("model_checkpoint_path:"):
            p = line.split(":", 1)[1].strip().strip('"')
            p = p if os.path.isabs(p) else os.path.join(checkpoint_dir, p)
            return p if os.path.exists(p + ".index") else None
    return None


# ------------------------------------------------------------------------------------------------------------ writer
def _pb_varint_field(f, v):
    return _put_varint((f << 3) | 0) + _put_varint(v)


def _pb_bytes_field(f, b):
    return _put_varint((f << 3) | 2) + _put_varint(len(b)) + b


def _table_block(entries, restart_interval=16):
    out, restarts, prev = bytearray(), [], b""
    for i, (k, v) in enumerate(entries):
        if i % restart_interval == 0:
            restarts.append(len(out))
            shared = 0
        else:
            shared = 0
            while shared < min(len(prev), len(k)) and prev[shared] == k[shared]:
                shared += 1
        out += _put_varint(shared) + _put_varint(len(k) - shared) + _put_varint(len(v)) + k[shared:] + v
        prev = k
    for r in restarts or [0]:
        out += struct.pack("<I", r)
    out += struct.pack("<I", max(len(restarts), 1))
    return bytes(out)


def write_checkpoint(prefix, tensors, block_size=4096, checksums=True):
    """Writes {name: numpy array} as a single-shard V2 checkpoint (uncompressed index blocks) + the 'checkpoint' state file.
    checksums=False leaves the per-tensor crc32c out (pure-python CRC is ~1 MB/s; TensorFlow itself would refuse such a file)."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    names = sorted(tensors, key=lambda s: s.encode("utf-8"))
    items = [(b"", _pb_varint_field(1, 1) + _pb_bytes_field(3, _pb_varint_field(1, 1)))]      # num_shards=1, version{producer=1}
    offset = 0
    with open(prefix + ".data-00000-of-00001", "wb") as f:
        for n in names:
            a = np.asarray(tensors[n])
            if not a.flags.c_contiguous:
                a = a.copy(order="C")
            if a.dtype not in DTYPE_ENUM:
                raise CheckpointError("%s: dtype %s not supported" % (n, a.dtype))
            raw = a.astype(a.dtype.newbyteorder("<")).tobytes()
            f.write(raw)
            shape = b"".join(_pb_bytes_field(2, _pb_varint_field(1, int(d))) for d in a.shape)
            e = _pb_varint_field(1, DTYPE_ENUM[a.dtype]) + _pb_bytes_field(2, shape)
            if offset:
                e += _pb_varint_field(4, offset)
            e += _pb_varint_field(5, len(raw))
            if checksums:
                e += _put_varint((6 << 3) | 5) + struct.pack("<I", mask_crc(crc32c(raw)))
            items.append((n.encode("utf-8"), e))
            offset += len(raw)
    out, index_entries, cur, cur_bytes = bytearray(), [], [], 0

    def flush():
        nonlocal cur, cur_bytes
        if not cur:
            return
        blk = _table_block(cur)
        handle = _put_varint(len(out)) + _put_varint(len(blk))
        out.extend(blk + b"\x00" + struct.pack("<I", mask_crc(crc32c(blk + b"\x00"))))
        index_entries.append((cur[-1][0], handle))
        cur, cur_bytes = [], 0

    for k, v in items:
        cur.append((k, v))
        cur_bytes += len(k) + len(v) + 8
        if cur_bytes >= block_size:
            flush()
    flush()
    meta = _table_block([])
    meta_handle = _put_varint(len(out)) + _put_varint(len(meta))
    out.extend(meta + b"\x00" + struct.pack("<I", mask_crc(crc32c(meta + b"\x00"))))
    idx = _table_block(index_entries, restart_interval=1)
    idx_handle = _put_varint(len(out)) + _put_varint(len(idx))
    out.extend(idx + b"\x00" + struct.pack("<I", mask_crc(crc32c(idx + b"\x00"))))
    footer = meta_handle + idx_handle
    out.extend(footer + b"\x00" * (40 - len(footer)) + struct.pack("<Q", TABLE_MAGIC))
    open(prefix + ".index", "wb").write(bytes(out))
    with open(os.path.join(os.path.dirname(os.path.abspath(prefix)), "checkpoint"), "w") as f:
        base = os.path.basename(prefix)
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (base, base))


# ------------------------------------------------------------------------------------------------------------ warm start
def map_variable_names(var_names, model_scope, checkpoint_model_scope, checkpoint_exclude_scopes=None, name_remap=None):
    """scaffolds.get_init_fn_for_scaffold's name rules (utility/scaffolds.py:27-58): {checkpoint name: variable name}.
    var_names are full graph names ('<model_scope>/conv1/conv1_1/conv2d/kernel')."""
    exclusions = [s.strip() for s in checkpoint_exclude_scopes.split(",")] if checkpoint_exclude_scopes else []
    kept = [v for v in var_names if not any(ex in v for ex in exclusions)]           # substring match, as the reference (:38)
    if checkpoint_model_scope is None:
        return {v: v for v in kept}
    if checkpoint_model_scope.strip() == "":
        mapped = {v.replace(model_scope + "/", ""): v for v in kept}
    else:
        mapped = {v.replace(model_scope, checkpoint_model_scope.strip()): v for v in kept}
    if name_remap is not None:
        renamed = {}
        for ck, v in mapped.items():
            for k, r in name_remap.items():
                if k in ck:
                    renamed[ck.replace(k, r)] = v
                    break
            else:
                renamed[ck] = v
        mapped = renamed
    return mapped


def init_from_checkpoint(variables, checkpoint_path, model_scope, checkpoint_model_scope, checkpoint_exclude_scopes=None,
                         ignore_missing_vars=False, name_remap=None):
    """Restores a VariableStore the way the reference's scaffold init_fn does (utility/scaffolds.py:24-88); returns the list of
    restored variable names.  `checkpoint_path` is a prefix or a directory holding a 'checkpoint' state file."""
    import torch
    if os.path.isdir(checkpoint_path):
        found = latest_checkpoint(checkpoint_path)
        if found is None:
            raise CheckpointError("no checkpoint in %s" % checkpoint_path)
        checkpoint_path = found
    reader = CheckpointReader(checkpoint_path)
    store_names = [n for n, _ in variables.named()]
    full = {model_scope + "/" + n: n for n in store_names}
    mapping = map_variable_names(list(full), model_scope, checkpoint_model_scope, checkpoint_exclude_scopes, name_remap)
    if not mapping:
        raise ValueError("variables_to_restore cannot be empty")                       # scaffolds.py:62-63
    restored, params = [], dict(variables.named())
    for ck_name, var_name in mapping.items():
        if not reader.has_tensor(ck_name):
            if ignore_missing_vars:
                continue
            raise CheckpointError("variable %s missing in checkpoint %s" % (ck_name, checkpoint_path))
        p = params[full[var_name]]
        t = reader.get_tensor(ck_name)
        if tuple(t.shape) != tuple(p.shape):                                           # Saver(reshape=False)
            raise CheckpointError("%s: checkpoint shape %s != variable shape %s" % (ck_name, tuple(t.shape), tuple(p.shape)))
        with torch.no_grad():
            p.copy_(torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32)).to(p.device))
        restored.append(full[var_name])
    _bump_weight_epoch()
    return restored


def _bump_weight_epoch():
    """Parameters were written through views (members of fused blocks keep their own version counters): the cached 16-bit packings of
    ops.packed_weights key on (WEIGHT_EPOCH, block version, pointer) and must be refreshed."""
    from .. import ops
    ops.WEIGHT_EPOCH += 1


def _momentum_views(trainer):
    """{variable name: view of the flat momentum buffer} — tf.train.MomentumOptimizer keeps one 'Momentum' slot per variable."""
    flat = trainer.flat
    out = {}
    for n, p in trainer.model.vs.named():
        off = p.data.data_ptr() - flat.w.data_ptr()
        assert off >= 0 and off % 4 == 0
        out[n] = flat.v.as_strided(p.data.shape, p.data.stride(), off // 4)
    return out


def save_checkpoint(variables, prefix, model_scope, checksums=False, trainer=None):
    """Writes the VariableStore under the reference's graph names ('<model_scope>/<name>'): trainable variables AND the non-trainable
    state the Estimator's Saver writes — batch-norm moving_mean / moving_variance (VariableStore.bufs) — and, given the trainer, the
    optimizer slots ('<var>/Momentum', tf.train.MomentumOptimizer train_dan.py:521) and 'global_step' (int64), so that a run resumed
    from its own checkpoint continues the learning-rate schedule and the momenta (train_dan.py:549-556 keeps 5 such checkpoints)."""
    tensors = {model_scope + "/" + n: p.detach().cpu().numpy() for n, p in variables.named()}
    for n, b in getattr(variables, "bufs", {}).items():
        tensors[model_scope + "/" + n] = b.detach().cpu().numpy()
    if trainer is not None:
        for n, v in _momentum_views(trainer).items():
            tensors[model_scope + "/" + n + "/Momentum"] = v.detach().cpu().contiguous().numpy()
        tensors["global_step"] = np.asarray(int(trainer.step_no), dtype=np.int64)
        if getattr(trainer, "ls_state", None) is not None:       # fp16 build: dynamic loss scale {scale, clean steps} (not a reference variable)
            tensors["danhip/loss_scale"] = trainer.ls_state[0:2].detach().cpu().numpy().astype(np.float32)
    write_checkpoint(prefix, tensors, checksums=checksums)


def restore_checkpoint(variables, checkpoint_path, model_scope, trainer=None, strict=True):
    """tf.train.Saver().restore of a checkpoint written by save_checkpoint (the Estimator's resume path; eval_dan.py:406-411): every
    variable, the batch-norm moving statistics and — given the trainer — the Momentum slots and global_step.  Returns the restored
    names.  strict: a variable / slot missing in the checkpoint is an error (Saver semantics); global_step and slots are optional when
    strict is False (e.g. an inference-only checkpoint)."""
    import torch
    if os.path.isdir(checkpoint_path):
        found = latest_checkpoint(checkpoint_path)
        if found is None:
            raise CheckpointError("no checkpoint in %s" % checkpoint_path)
        checkpoint_path = found
    reader = CheckpointReader(checkpoint_path)
    restored = []

    def load(ck_name, dst):
        if not reader.has_tensor(ck_name):
            if strict:
                raise CheckpointError("variable %s missing in checkpoint %s" % (ck_name, checkpoint_path))
            return
        t = reader.get_tensor(ck_name)
        if tuple(t.shape) != tuple(dst.shape):
            raise CheckpointError("%s: checkpoint shape %s != variable shape %s" % (ck_name, tuple(t.shape), tuple(dst.shape)))
        with torch.no_grad():
            dst.copy_(torch.from_numpy(np.ascontiguousarray(t, dtype=np.float32)).to(dst.device))
        restored.append(ck_name)

    for n, p in variables.named():
        load(model_scope + "/" + n, p)           # (under no_grad, on the Parameter itself: its version counter moves, as init_from_checkpoint does)
    for n, b in getattr(variables, "bufs", {}).items():
        load(model_scope + "/" + n, b)
    if trainer is not None:
        for n, v in _momentum_views(trainer).items():
            load(model_scope + "/" + n + "/Momentum", v)
        if reader.has_tensor("global_step"):
            trainer.step_no = int(reader.get_tensor("global_step"))
            restored.append("global_step")
        elif strict:
            raise CheckpointError("global_step missing in checkpoint %s" % checkpoint_path)
        if getattr(trainer, "ls_state", None) is not None and reader.has_tensor("danhip/loss_scale"):      # optional: reference checkpoints lack it
            ls = np.asarray(reader.get_tensor("danhip/loss_scale"), dtype=np.float32).reshape(-1)
            with torch.no_grad():
                trainer.ls_state[0:2].copy_(torch.from_numpy(ls[:2].copy()).to(trainer.ls_state.device))
            restored.append("danhip/loss_scale")
        if trainer._graph is not None:           # a captured step holds the old learning rate: the next train_step re-captures
            trainer._graph = None
            trainer._recapture = True
    _bump_weight_epoch()
    return restored
