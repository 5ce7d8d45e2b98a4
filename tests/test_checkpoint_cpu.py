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
