"""TF V2 checkpoint reader / writer and the scaffold warm-start rules (utility/scaffolds.py:24-88) — host code, CPU only.
The reference ships no checkpoint fixture and TensorFlow is absent, so the format is pinned by its published constants
(table magic, CRC-32C check value, snappy / varint / prefix-compression rules) and by write -> read round trips."""
import os
import struct

import numpy as np
import pytest
import torch

from dan_amd.utility import checkpoint as C


def test_crc32c_and_mask_known_answers():
    assert C.crc32c(b"123456789") == 0xE3069283                     # the CRC-32C check value
    assert C.crc32c(b"") == 0
    assert C.mask_crc(0) == 0xa282ead8
    assert C.crc32c(b"6789", C.crc32c(b"12345")) == 0xE3069283      # incremental form


def test_snappy_decompress_literal_and_overlapping_copy():
    # "abcdabcdabcdab": literal "abcd" (tag (4-1)<<2), then a 10-byte copy at offset 4 (kind 2: ((10-1)<<2)|2, offset LE16)
    stream = bytes([14]) + bytes([(3 << 2) | 0]) + b"abcd" + bytes([((10 - 1) << 2) | 2, 4, 0])
    assert C.snappy_decompress(stream) == b"abcdabcdabcdab"
    with pytest.raises(C.CheckpointError):
        C.snappy_decompress(bytes([5]) + bytes([(3 << 2) | 0]) + b"abcd")


def test_round_trip_many_variables_several_blocks(tmp_path):
    rng = np.random.RandomState(0)
    tensors = {"sfd/conv%d/conv%d_%d/conv2d/%s" % (i, i, j, k): rng.randn(*shape).astype(np.float32)
               for i in range(1, 8) for j in range(1, 6) for k, shape in (("kernel", (3, 3, 4, 5)), ("bias", (5,)))}
    tensors["global_step"] = np.array(12345, dtype=np.int64)
    tensors["sfd/l2_norm_layer_3/weight"] = np.full((256,), 10.0, dtype=np.float32)
    tensors["half"] = rng.randn(3, 2).astype(np.float16)
    prefix = str(tmp_path / "ck" / "model.ckpt-7")
    C.write_checkpoint(prefix, tensors, block_size=512)              # small blocks: several data blocks + prefix compression
    data = open(prefix + ".index", "rb").read()
    assert struct.unpack("<Q", data[-8:])[0] == 0xdb4775248b80fb57
    r = C.CheckpointReader(prefix)
    assert set(r.get_variable_to_shape_map()) == set(tensors)
    assert r.get_variable_to_shape_map()["global_step"] == [] and r.get_variable_to_dtype_map()["global_step"] == np.int64
    for n, t in tensors.items():
        got = r.get_tensor(n, verify=True)
        assert got.dtype == t.dtype and got.shape == t.shape and np.array_equal(got, t), n
    assert not r.has_tensor("sfd/conv9/nope")
    with pytest.raises(KeyError):
        r.get_tensor("sfd/conv9/nope")
    assert C.latest_checkpoint(str(tmp_path / "ck")) == prefix
    assert C.latest_checkpoint(str(tmp_path)) is None
    # corruption is detected: flip one byte inside the first data block / inside a tensor
    bad = bytearray(data); bad[10] ^= 0xFF
    open(prefix + ".index", "wb").write(bytes(bad))
    with pytest.raises(C.CheckpointError):
        C.CheckpointReader(prefix)
    open(prefix + ".index", "wb").write(data)
    raw = bytearray(open(prefix + ".data-00000-of-00001", "rb").read()); raw[3] ^= 0x55
    open(prefix + ".data-00000-of-00001", "wb").write(bytes(raw))
    first = sorted(tensors, key=lambda s: s.encode())[0]
    with pytest.raises(C.CheckpointError):
        C.CheckpointReader(prefix).get_tensor(first, verify=True)


def test_name_rules_of_the_scaffold():
    names = ["sfd/conv1/conv1_1/conv2d/kernel", "sfd/conv1/conv1_1/conv2d/bias", "sfd/l2_norm_layer_3/weight", "sfd/multibox_head/loc_0/kernel",
             "sfd/additional_layers/conv6_1/conv2d/kernel"]
    remap = {"/conv2d/kernel": "/weights", "/conv2d/bias": "/biases"}                 # train_sfd.py:163
    m = C.map_variable_names(names, "sfd", "vgg_16", "sfd/multibox_head, sfd/additional_layers,sfd/l2_norm", remap)
    assert m == {"vgg_16/conv1/conv1_1/weights": "sfd/conv1/conv1_1/conv2d/kernel", "vgg_16/conv1/conv1_1/biases": "sfd/conv1/conv1_1/conv2d/bias"}
    m = C.map_variable_names(names[:2], "sfd", "", None, None)                        # empty checkpoint scope strips '<scope>/'
    assert set(m) == {"conv1/conv1_1/conv2d/kernel", "conv1/conv1_1/conv2d/bias"}
    assert C.map_variable_names(names[:1], "sfd", None) == {names[0]: names[0]}


def test_warm_start_into_a_variable_store(tmp_path):
    from dan_amd.net.variables import VariableStore
    src = VariableStore(device="cpu", seed=1)
    src.get("conv1/conv1_1/conv2d/kernel", (3, 3, 3, 8), "glorot")
    src.get("conv1/conv1_1/conv2d/bias", (8,), 0.25)
    src.get("l2_norm_layer_3/weight", (8,), 10.0)
    prefix = str(tmp_path / "pre" / "vgg.ckpt")
    # a "pretrained VGG": slim names under scope vgg_16, no l2-norm weights, conv kernels HWIO as in TF
    C.write_checkpoint(prefix, {"vgg_16/conv1/conv1_1/weights": src.vars[src._key("conv1/conv1_1/conv2d/kernel")].detach().numpy(),
                                "vgg_16/conv1/conv1_1/biases": src.vars[src._key("conv1/conv1_1/conv2d/bias")].detach().numpy()})
    dst = VariableStore(device="cpu", seed=2)
    k = dst.get("conv1/conv1_1/conv2d/kernel", (3, 3, 3, 8), "glorot")
    b = dst.get("conv1/conv1_1/conv2d/bias", (8,), "zeros")
    g = dst.get("l2_norm_layer_3/weight", (8,), 10.0)
    remap = {"/conv2d/kernel": "/weights", "/conv2d/bias": "/biases"}
    with pytest.raises(C.CheckpointError):                                           # l2-norm weight is not in the checkpoint
        C.init_from_checkpoint(dst, prefix, "sfd", "vgg_16", None, False, remap)
    got = C.init_from_checkpoint(dst, str(tmp_path / "pre"), "sfd", "vgg_16", None, True, remap)   # directory + ignore_missing_vars
    assert sorted(got) == ["conv1/conv1_1/conv2d/bias", "conv1/conv1_1/conv2d/kernel"]
    assert torch.equal(k, src.vars[src._key("conv1/conv1_1/conv2d/kernel")]) and torch.equal(b, torch.full((8,), 0.25))
    assert torch.equal(g, torch.full((8,), 10.0))
    # shape mismatch is an error (Saver(reshape=False)); own checkpoints round-trip under '<scope>/<name>'
    bad = VariableStore(device="cpu", seed=3)
    bad.get("conv1/conv1_1/conv2d/kernel", (3, 3, 3, 16), "glorot")
    with pytest.raises(C.CheckpointError):
        C.init_from_checkpoint(bad, prefix, "sfd", "vgg_16", None, True, remap)
    own = str(tmp_path / "own" / "model.ckpt-1")
    C.save_checkpoint(dst, own, "sfd")
    again = VariableStore(device="cpu", seed=4)
    for n, p in dst.named():
        again.get(n, p.shape, "zeros")
    assert len(C.init_from_checkpoint(again, own, "sfd", None)) == 3
    for (n, p), (_, q) in zip(dst.named(), again.named()):
        assert torch.equal(p, q), n


def test_full_state_round_trip_with_momenta_moving_statistics_and_global_step(tmp_path):
    """save_checkpoint(trainer=...) / restore_checkpoint: what the Estimator's Saver writes besides the trainable variables — batch-norm
    moving statistics, '<var>/Momentum' slots, 'global_step' (int64) — survives a round trip into a fresh store whose head variables live
    as strided views of a fused block; the cached weight packings are invalidated (ops.WEIGHT_EPOCH)."""
    from dan_amd import ops
    from dan_amd.net.variables import VariableStore
    from dan_amd.trainer import FlatParams

    class FakeModel(object):
        pass

    class FakeTrainer(object):
        _graph = None

    def make(seed):
        vs = VariableStore(device="cpu", seed=seed)
        vs.get("conv1/conv1_1/conv2d/kernel", (3, 3, 8, 8), "glorot")
        vs.get("conv1/conv1_1/conv2d/bias", (8,), 0.1 * seed)
        vs.get("h/loc_0/kernel", (3, 3, 8, 4), "glorot")
        vs.get("h/loc_0/bias", (4,), "zeros")
        vs.get("h/cls_0/kernel", (3, 3, 8, 2), "glorot")
        vs.get("h/cls_0/bias", (2,), 0.5)
        vs.buffer("block_0/conv_1/bn/moving_mean", (8,), 0.0).add_(float(seed))
        vs.buffer("block_0/conv_1/bn/moving_variance", (8,), 1.0).mul_(float(seed))
        vs.fuse(("h/loc_0/kernel", "h/cls_0/kernel"), 3)
        vs.fuse(("h/loc_0/bias", "h/cls_0/bias"), 0)
        t = FakeTrainer()
        t.model = FakeModel()
        t.model.vs = vs
        t.flat = FlatParams(vs)
        t.step_no = 0
        return t

    a = make(3)
    a.flat.v.copy_(torch.arange(a.flat.total, dtype=torch.float32) * 1e-3)
    a.step_no = 81234
    prefix = str(tmp_path / "run" / "model.ckpt-81234")
    C.save_checkpoint(a.model.vs, prefix, "dan", trainer=a)
    names = C.CheckpointReader(prefix).get_variable_to_shape_map()
    assert "global_step" in names and "dan/h/cls_0/kernel/Momentum" in names and "dan/block_0/conv_1/bn/moving_variance" in names
    assert names["dan/h/cls_0/kernel/Momentum"] == [3, 3, 8, 2] and names["global_step"] == []
    b = make(5)
    epoch = ops.WEIGHT_EPOCH
    got = C.restore_checkpoint(b.model.vs, str(tmp_path / "run"), "dan", trainer=b)
    assert ops.WEIGHT_EPOCH > epoch and "global_step" in got and b.step_no == 81234
    for (n, p), (_, q) in zip(a.model.vs.named(), b.model.vs.named()):
        assert torch.equal(p, q), n
    for n in a.model.vs.bufs:
        assert torch.equal(a.model.vs.bufs[n], b.model.vs.bufs[n]), n
    # momenta: compare per variable through the trainer's own views (the flat buffers' padding is not saved)
    va, vb = C._momentum_views(a), C._momentum_views(b)
    assert all(torch.equal(va[n], vb[n]) for n in va) and vb["h/cls_0/kernel"].abs().sum() > 0
    # an inference checkpoint (no slots, no step) restores strictly into a plain store, and non-strictly into a trainer
    inf = str(tmp_path / "inf" / "model.ckpt")
    C.save_checkpoint(a.model.vs, inf, "dan")
    with pytest.raises(C.CheckpointError):
        C.restore_checkpoint(b.model.vs, inf, "dan", trainer=b)
    assert "global_step" not in C.restore_checkpoint(b.model.vs, inf, "dan", trainer=b, strict=False)
    epoch = ops.WEIGHT_EPOCH
    C.init_from_checkpoint(b.model.vs, inf, "dan", None)
    assert ops.WEIGHT_EPOCH > epoch                                   # warm start also invalidates the cached packings


def test_fused_inference_blocks_follow_a_restore_and_raw_pointer_writers(tmp_path):
    """the trainer-less evaluation flow — a gradient-free forward creates and fuses the variables, THEN the checkpoint is
    restored — must see the restored weights in every fused block (heads, context 'plus' / block-diagonal kernels, the deformable HWIO
    operand).  Also the writers that bump no version counter: `p.data.copy_()` and anything that advances ops.WEIGHT_EPOCH."""
    from dan_amd import ops
    from dan_amd.net.variables import VariableStore

    def make(seed):
        vs = VariableStore(device="cpu", seed=seed)
        vs.get("h/loc_0/kernel", (3, 3, 8, 4), "glorot")
        vs.get("h/cls_0/kernel", (3, 3, 8, 2), "glorot")
        vs.get("b/c31/kernel", (3, 1, 8, 4), "glorot")
        vs.get("b/c13/kernel", (1, 3, 8, 4), "glorot")
        vs.get("d/deform_conv/kernel", (8, 4, 3, 3), "glorot_oihw")
        return vs

    groups = [(("h/loc_0/kernel", "h/cls_0/kernel"), 3), (("b/c31/kernel", "b/c13/kernel"), "plus"), (("d/deform_conv/kernel",), "hwio")]
    src = make(11)
    prefix = str(tmp_path / "m" / "model.ckpt-7")
    C.save_checkpoint(src, prefix, "dan")
    dst = make(12)
    with torch.no_grad():                                    # the forward that precedes the restore
        before = [dst.fuse(k, a) for k, a in groups]
    assert all(t is not None for t in before)
    C.restore_checkpoint(dst, prefix, "dan", trainer=None, strict=False)
    with torch.no_grad():
        after = [dst.fuse(k, a) for k, a in groups]
        want = [src.build(k, a) for k, a in groups]
    for (k, _), t0, t1, w in zip(groups, before, after, want):
        assert t1 is not t0, k                               # a new block, not the pre-restore one
        assert torch.equal(t1, w), k                         # ... that matches its (restored) members
    # a write through .data (its own version counter) is invisible to p._version: the epoch is what the cache keys on
    with torch.no_grad():
        t_old = dst.fuse(*groups[0])
        p = dst.vars[dst._key("h/cls_0/kernel")]
        v0 = p._version
        p.data.mul_(2.0)
        assert p._version == v0 and dst.fuse(*groups[0]) is t_old          # (documented blind spot of the version counter ...)
        ops.WEIGHT_EPOCH += 1                                              # ... which every raw-pointer writer closes this way
        t_new = dst.fuse(*groups[0])
        assert t_new is not t_old and torch.equal(t_new[..., 4:], p)


def test_reader_accepts_a_bundle_written_by_an_independent_encoder(tmp_path):
    """SURVEY 8f row 2 / VERDICT r5 (f2 "partial": the reader had only ever read its own writer's files).  tests/tf_bundle_encoder.py writes
    the TF-V2 format from the format definitions with different free choices (two shards, restart interval 1, small blocks, a CRC per
    tensor, an empty-shape int64 scalar): every tensor comes back bit for bit, checksums verified, shapes / dtypes as written."""
    import numpy as np
    from tf_bundle_encoder import crc32c as crc_indep, write_bundle
    from dan_amd.utility import checkpoint as C
    assert crc_indep(b"123456789") == 0xE3069283 and C.crc32c(b"123456789") == 0xE3069283        # the CRC-32C check value, both implementations
    rng = np.random.default_rng(3)
    tensors = {"vgg_16/conv1/conv1_%d/weights" % i: rng.standard_normal((3, 3, 8 * i + 3, 16), dtype=np.float32) for i in range(1, 9)}
    tensors.update({"vgg_16/conv1/conv1_%d/biases" % i: rng.standard_normal((16,), dtype=np.float32) for i in range(1, 9)})
    tensors["global_step"] = np.asarray(120000, dtype=np.int64)
    tensors["a/very/" + "long/" * 40 + "name"] = np.arange(7, dtype=np.int32)
    prefix = str(tmp_path / "model.ckpt-120000")
    write_bundle(prefix, tensors)
    r = C.CheckpointReader(prefix)
    assert r.num_shards == 2 and set(r.get_variable_to_shape_map()) == set(tensors)
    for n, a in tensors.items():
        got = r.get_tensor(n, verify=True)
        assert got.dtype == a.dtype and got.shape == a.shape and np.array_equal(got, a), n
    assert int(r.get_tensor("global_step")) == 120000
