"""Bit-exact parity of the DynamicAnchorRouting and NMS kernels (through the C ABI / the custom_op + bbox_util mirrors)
against the oracle's C++ / numpy restatements, incl. the golden KAT, order-dependent corner cases (easy-background cells
claimed before / after their own turn, ties), degenerate boxes and the seeded reservoir stream."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import anchors as OA
from oracle import extra_lib as OX

pytestmark = pytest.mark.gpu
KATS = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kats.json")))


def _rand_case(seed, fh, fw, depth, stride, n_hot=None):
    rng = np.random.RandomState(seed)
    N = fh * fw * depth
    cy = (rng.rand(N) * fh * stride).astype(np.float32)
    cx = (rng.rand(N) * fw * stride).astype(np.float32)
    h = (rng.rand(N) * 6 * stride).astype(np.float32)
    w = (rng.rand(N) * 6 * stride).astype(np.float32)
    h[rng.rand(N) < 0.05] = 0.3                                   # degenerate boxes (< 1 px)
    anchors = np.stack([cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2], -1).astype(np.float32)
    anchors[rng.rand(N) < 0.03] += np.float32(3 * fh * stride)    # far outside the map
    labels = rng.rand(N).astype(np.float32)
    labels[rng.rand(N) < 0.1] = 0.0
    labels = np.round(labels * 8) / 8 if seed % 2 else labels     # many exact ties on odd seeds
    mask_in = (rng.rand(N) < 0.6).astype(np.int32)
    gt = (rng.randn(N, 4) * 0.2).astype(np.float32)
    return anchors, gt, labels.astype(np.float32), mask_in


def test_routing_eval_kat(dev):
    from dan_amd.utility import custom_op
    k = KATS["routing_eval"]
    t = lambda a, dt: torch.tensor(a, dtype=dt, device=dev)
    mo, do = custom_op.dynamic_anchor_routing(t(k["anchors"], torch.float32), torch.zeros((4, 4), device=dev), t(k["labels"], torch.float32),
                                              t(k["mask_in"], torch.int32), k["feat"][0], k["feat"][1], k["depth"], k["stride"], 8, 8, False, 0.03, 0.0)
    assert mo.cpu().tolist() == k["mask_out"]
    assert do.cpu().tolist() == k["decode_out"]


def test_routing_train_kat(dev):
    """The hand-traced train-mode KAT (dynamic_anchor_routing.cc:203-327; kats.json "routing_train") on the HIP kernels: the library draws
    from its counter-based stream, and the recorded (seed, counter0) takes the traced accept / reject decisions (tests/test_oracle_cpu.py)."""
    from dan_amd.utility import custom_op
    k = KATS["routing_train"]
    t = lambda key, dt: torch.tensor(k[key], dtype=dt, device=dev)
    mo, do = custom_op.dynamic_anchor_routing(t("anchors", torch.float32), t("gt", torch.float32), t("labels", torch.float32), t("mask_in", torch.int32),
                                              k["feat"][0], k["feat"][1], k["depth"], k["stride"], 8, 12, True, k["thres"], k["ignore_thres"],
                                              seed=k["stream"]["seed"], counter0=k["stream"]["counter0"])
    assert mo.cpu().tolist() == k["mask_out"]
    assert np.allclose(do.cpu().numpy(), np.asarray(k["decode_out"], np.float32), rtol=3e-7, atol=0)      # log() from the device libm


@pytest.mark.parametrize("seed,fh,fw,depth,stride", [(0, 8, 8, 1, 4), (1, 16, 12, 1, 8), (2, 5, 7, 2, 16), (3, 40, 40, 1, 4), (5, 3, 3, 3, 32)])
def test_routing_eval_matches_oracle(seed, fh, fw, depth, stride, dev):
    from dan_amd.utility import custom_op
    anchors, gt, labels, mask_in = _rand_case(seed, fh, fw, depth, stride)
    mo_ref, do_ref = OX.dynamic_anchor_routing(anchors, gt, labels, mask_in, fh, fw, depth, stride, 0, 0, False, 0.0, 0.0)
    t = lambda a: torch.from_numpy(a).to(dev)
    mo, do = custom_op.dynamic_anchor_routing(t(anchors), t(gt), t(labels), t(mask_in), fh, fw, depth, stride, 0, 0, False, 0.03, 0.0)
    assert np.array_equal(mo.cpu().numpy(), mo_ref)
    got = do.cpu().numpy()
    # exp() comes from the device libm: allow 2 ulp on the decoded coordinates, exact elsewhere
    assert np.allclose(got, do_ref, rtol=3e-7, atol=1e-5), np.abs(got - do_ref).max()


def test_routing_eval_batched(dev):
    from dan_amd.utility import custom_op
    cases = [_rand_case(s, 10, 10, 1, 8) for s in (10, 11, 12)]
    st = lambda j: torch.from_numpy(np.stack([c[j] for c in cases])).to(dev)
    mo, do = custom_op.dynamic_anchor_routing(st(0), st(1), st(2), st(3), 10, 10, 1, 8, 0, 0, False, 0.03, 0.0)
    for b, c in enumerate(cases):
        mo_ref, do_ref = OX.dynamic_anchor_routing(c[0], c[1], c[2], c[3], 10, 10, 1, 8, 0, 0, False, 0.0, 0.0)
        assert np.array_equal(mo[b].cpu().numpy(), mo_ref)
        assert np.allclose(do[b].cpu().numpy(), do_ref, rtol=3e-7, atol=1e-5)


@pytest.mark.parametrize("seed,fh,fw,depth,stride", [(0, 8, 8, 1, 4), (1, 16, 12, 1, 8), (2, 5, 7, 2, 16), (3, 40, 40, 1, 4)])
def test_routing_train_matches_oracle(seed, fh, fw, depth, stride, dev):
    from dan_amd.utility import custom_op
    rng = np.random.RandomState(100 + seed)
    N = fh * fw * depth
    anchors, _, labels, mask_in = _rand_case(seed, fh, fw, depth, stride)
    # gt boxes (absolute) near the anchors so that IoUs are meaningful; labels > 0 mark stage-1 positives
    jitter = (rng.randn(N, 4) * stride * 0.5).astype(np.float32)
    gt = (anchors + jitter).astype(np.float32)
    labels = (rng.rand(N) < 0.4).astype(np.float32)
    sd, c0 = 1234 + seed, 77
    u = OX.uniform_stream(sd, c0, N)
    mo_ref, do_ref = OX.dynamic_anchor_routing(anchors, gt, labels, mask_in, fh, fw, depth, stride, 0, 0, True, 0.5, 0.35, u=u)
    t = lambda a: torch.from_numpy(a).to(dev)
    mo, do = custom_op.dynamic_anchor_routing(t(anchors), t(gt), t(labels), t(mask_in), fh, fw, depth, stride, 0, 0, True, 0.5, 0.35, seed=sd, counter0=c0)
    assert np.array_equal(mo.cpu().numpy(), mo_ref)
    got = do.cpu().numpy()
    assert np.allclose(got, do_ref, rtol=3e-7, atol=1e-6), np.abs(got - do_ref).max()      # log() from the device libm
    assert (mo_ref == 1).sum() > 0 and (mo_ref == -1).sum() > 0


def test_routing_rejects_bad_arguments(dev):
    from dan_amd.utility import custom_op
    z = torch.zeros((4, 4), device=dev)
    with pytest.raises(ValueError):
        custom_op.dynamic_anchor_routing(z, z, torch.zeros(4, device=dev), torch.zeros(4, dtype=torch.int32, device=dev), 2, 2, 1, 4, 8, 8, True, 1.0, 0.0)
    with pytest.raises(ValueError):
        custom_op.dynamic_anchor_routing(z, z, torch.zeros(4, device=dev), torch.zeros(4, dtype=torch.int32, device=dev), 3, 2, 1, 4, 8, 8, False, 0.5, 0.0)


@pytest.mark.parametrize("seed,K,topk,thr", [(0, 50, 20, 0.3), (1, 400, 200, 0.45), (2, 1125, 750, 0.3), (3, 7, 10, 0.0)])
def test_nms_matches_oracle(seed, K, topk, thr, dev):
    from dan_amd.utility import bbox_util
    rng = np.random.RandomState(seed)
    cy, cx = rng.rand(K) * 200, rng.rand(K) * 200
    h, w = rng.rand(K) * 60 + 1, rng.rand(K) * 60 + 1
    boxes = np.stack([cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2], -1).astype(np.float32)
    boxes[::17] = 0.0                                               # zero-area padding rows, as sort_bboxes produces
    scores = (np.round(rng.rand(K) * 50) / 50).astype(np.float32)   # ties
    scores[::17] = 0.0
    keep = OA.nms_tf(boxes, scores, topk, thr)
    s, b = bbox_util.nms_bboxes(torch.from_numpy(scores).to(dev), torch.from_numpy(boxes).to(dev), topk, thr)
    assert np.array_equal(s.cpu().numpy(), scores[keep])
    assert np.array_equal(b.cpu().numpy(), boxes[keep])
    sp, bp = bbox_util.nms_bboxes_with_padding(torch.from_numpy(scores).to(dev), torch.from_numpy(boxes).to(dev), topk, thr)
    assert sp.shape == (topk,) and bp.shape == (topk, 4)
    assert np.array_equal(sp.cpu().numpy()[:len(keep)], scores[keep]) and float(sp[len(keep):].abs().sum()) == 0.0


def test_parse_by_class_matches_oracle(dev):
    from dan_amd.utility import bbox_util
    rng = np.random.RandomState(7)
    A = 600
    logits = (rng.randn(A, 2) * 3).astype(np.float32)
    cy, cx = rng.rand(A) * 300, rng.rand(A) * 300
    h, w = rng.rand(A) * 80, rng.rand(A) * 80
    boxes = np.stack([cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2], -1).astype(np.float32)
    ob, osc = OA.parse_by_class(logits, boxes, (300, 300), 0.2, 4.0, 200, 100, 0.3)
    bb, sc = bbox_util.parse_by_class((300, 300), torch.from_numpy(logits).to(dev), torch.from_numpy(boxes).to(dev), 2, 0.2, 4.0, 200, 100, 0.3)
    assert np.allclose(sc[1].cpu().numpy(), osc, atol=2e-6)          # softmax: device exp
    assert np.allclose(bb[1].cpu().numpy(), ob, atol=1e-4)
