"""Bit-exact parity of the HIP anchor / matching / encode / decode kernels (through the C ABI) with the numpy + C++ oracle."""
import numpy as np
import pytest
import torch

from oracle import anchors as OA
from oracle import extra_lib as OE

pytestmark = pytest.mark.gpu


def _cfg(h, w, ratio):
    from dan_amd.train_sfd import ALL_ANCHOR_SCALES, ALL_LAYER_STRIDES, layer_shapes
    shapes = layer_shapes(h, w)
    hs, ws, ds = [], [], []
    for i in range(6):
        a, b, d = OA.get_anchors_width_height(ALL_ANCHOR_SCALES[i], (), (ratio,))
        hs.append(a); ws.append(b); ds.append(d)
    return OA.get_all_anchors((h, w), hs, ws, ds, [0.5] * 6, shapes, ALL_LAYER_STRIDES, [float(h)] * 6, [False] * 6)


def _gt(n, h, w, seed):
    rng = np.random.RandomState(seed)
    side = np.exp(rng.uniform(np.log(8), np.log(min(h, w) * 0.6), n))
    cy, cx = rng.uniform(0, h, n), rng.uniform(0, w, n)
    b = np.stack([cy - side / 2, cx - side / 2, cy + side / 2, cx + side / 2], -1)
    b = np.round(np.clip(b, 0, [h - 1, w - 1, h - 1, w - 1])).astype(np.float32)
    return b[(b[:, 2] - b[:, 0] > 3) & (b[:, 3] - b[:, 1] > 3)]


@pytest.mark.parametrize("hw,ratio", [((640, 640), 1.0), ((640, 640), 0.8), ((320, 448), 1.0), ((101, 75), 0.8)])
def test_anchor_generation_bit_exact(hw, ratio, dev):
    from dan_amd.train_sfd import AnchorConfig
    ref = _cfg(hw[0], hw[1], ratio)
    cfg = AnchorConfig(hw[0], hw[1], dev, ratios=[(ratio,)] * 6)
    for got, want in zip(cfg.anchors[:4], ref[:4]):
        assert np.array_equal(got.cpu().numpy(), want)
    assert np.array_equal(cfg.anchors[4].cpu().numpy(), ref[4])
    if hw == (640, 640):
        assert cfg.num_anchors == 34125
        if ratio == 1.0:   # KAT
            assert [float(a[0]) for a in cfg.anchors[:4]] == [-5.5, -5.5, 9.5, 9.5]


@pytest.mark.parametrize("seed,n", [(0, 1), (1, 7), (2, 40), (3, 3)])
def test_iou_and_matching_bit_exact(seed, n, dev):
    from dan_amd.train_sfd import AnchorConfig
    from dan_amd.utility import anchor_manipulator as AM
    h = w = 320
    ref = _cfg(h, w, 1.0)
    cfg = AnchorConfig(h, w, dev)
    gt = _gt(n, h, w, seed)
    if gt.shape[0] == 0:
        gt = np.asarray([[0, 0, 1, 1]], np.float32)
    all_a = np.stack(ref[:4], -1)
    ov_ref = OA.iou_matrix(all_a, gt) * ref[4].astype(np.float32)[:, None]
    ov = AM.iou_matrix(cfg.anchors[:4], torch.from_numpy(gt).to(dev), cfg.anchors[4])
    assert np.array_equal(ov.cpu().numpy(), ov_ref)
    # small-mining match (S3FD / PB face): thresholds of train_sfd.py
    i_ref, s_ref = OE.small_mining_match(ov_ref, 0., 0.4, 0.4, 6, 0.3)
    i_got, s_got = AM.small_mining_match(ov, 0., 0.4, 0.4, 6, 0.3)
    assert np.array_equal(i_got.cpu().numpy(), i_ref)
    assert np.array_equal(s_got.cpu().numpy(), s_ref)
    # dual-max match (DAN): 0.35 / 0.35
    d_ref, ds_ref = OA.do_dual_max_match(ov_ref, 0.35, 0.35)
    d_got, ds_got = AM.do_dual_max_match(ov, 0.35, 0.35)
    assert np.array_equal(d_got.cpu().numpy().astype(np.int64), d_ref)
    assert np.array_equal(ds_got.cpu().numpy(), ds_ref)


def test_small_mining_match_kats_and_ties(dev):
    """KAT 1 of SURVEY 8(c) (inputs from cpp/ExtraLib/test_op.py:41,56) + tie-heavy random matrices that exercise the
    priority-queue order of the hard-face compensation phase."""
    from dan_amd.utility import anchor_manipulator as AM
    ov = np.array([[0.1, 0.4, 0.6, 0.2, 0.7], [0.5, 0.14, 0.76, 0.32, 0.47], [0.21, 0.94, 0.66, 0.22, 0.57],
                   [0.91, 0.14, 0.26, 0.42, 0.67], [0.11, 0.84, 0.26, 0.42, 0.57]], np.float32)
    for args in ((0., 0.6, 0.6, 5, 0.1), (0., 0.5, 0.5, 2, 0.1)):
        i, s = AM.small_mining_match(torch.from_numpy(ov).to(dev), *args)
        assert i.cpu().tolist() == [4, 2, 1, 3, 3]
        assert np.array_equal(s.cpu().numpy(), np.array([0.7, 0.76, 0.94, 0.42, 0.42], np.float32))
    # the hand-traced phase-3 KAT (kats.json "small_mining_match_phase3": equal IoUs in the heap, min_match binding, a stolen candidate)
    import json
    import os
    k = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "kats.json")))["small_mining_match_phase3"]
    i, s = AM.small_mining_match(torch.tensor(k["overlaps"], dtype=torch.float32, device=dev), *k["call"])
    assert i.cpu().tolist() == k["match_indices"]
    assert np.array_equal(s.cpu().numpy(), np.asarray(k["match_scores"], np.float32))
    rng = np.random.RandomState(11)
    for trial in range(12):
        A, G = int(rng.randint(50, 3000)), int(rng.randint(1, 30))
        levels = np.asarray([0.0, 0.05, 0.31, 0.32, 0.35, 0.38, 0.41, 0.5, 0.7], np.float32)
        ovr = levels[rng.randint(0, len(levels), (A, G))]          # many exact ties
        ovr[rng.rand(A, G) < 0.7] = 0.0
        i_ref, s_ref = OE.small_mining_match(ovr, 0., 0.4, 0.4, 6, 0.3)
        i_got, s_got = AM.small_mining_match(torch.from_numpy(ovr).to(dev), 0., 0.4, 0.4, 6, 0.3)
        assert np.array_equal(i_got.cpu().numpy(), i_ref), trial
        assert np.array_equal(s_got.cpu().numpy(), s_ref), trial


def test_dual_max_zero_column_quirk(dev):
    """A gt overlapping no anchor (column max == 0) force-matches every zero-overlap anchor."""
    from dan_amd.utility import anchor_manipulator as AM
    ov = np.zeros((64, 3), np.float32)
    ov[5, 0] = 0.5; ov[9, 2] = 0.7; ov[9, 0] = 0.2
    r_i, r_s = OA.do_dual_max_match(ov, 0.35, 0.35)
    g_i, g_s = AM.do_dual_max_match(torch.from_numpy(ov).to(dev), 0.35, 0.35)
    assert np.array_equal(g_i.cpu().numpy().astype(np.int64), r_i) and np.array_equal(g_s.cpu().numpy(), r_s)


@pytest.mark.parametrize("mining", [True, False])
def test_encode_decode(mining, dev):
    from dan_amd.train_sfd import AnchorConfig
    h = w = 320
    ref = _cfg(h, w, 1.0)
    cfg = AnchorConfig(h, w, dev, match_threshold=0.4 if mining else 0.35, neg_threshold=0.4 if mining else 0.35)
    gt = _gt(12, h, w, 5)
    mf = (lambda ov: OE.small_mining_match(ov, 0., 0.4, 0.4, 6, 0.3)) if mining else (lambda ov: OA.do_dual_max_match(ov, 0.35, 0.35))
    t_ref, l_ref, s_ref, m_ref = OA.encode_anchors(gt, ref[:4], ref[4], 0.4, 0.4, [0.1, 0.1, 0.2, 0.2], mf)
    t, l, s, m = cfg.enc.encode_anchors(torch.from_numpy(gt).to(dev), *cfg.anchors, match_mining=mining)
    assert np.array_equal(l.cpu().numpy().astype(np.int64), l_ref)
    assert np.array_equal(s.cpu().numpy(), s_ref)
    assert np.array_equal(m.cpu().numpy(), m_ref)
    tg = t.cpu().numpy()
    assert np.array_equal(tg[:, :2], t_ref[:, :2])                     # centre offsets: +,-,/ only -> bit exact
    # log(): device logf vs numpy logf may differ in the last ulp; declared tolerance 2 ulp of the log value / 0.2
    assert np.allclose(tg[:, 2:], t_ref[:, 2:], rtol=0, atol=4 * np.finfo(np.float32).eps * 5 * 4)
    # decode: zero offsets return the anchors themselves (exp(0) = 1 exactly)
    z = torch.zeros((2, cfg.num_anchors, 4), device=dev)
    d0 = cfg.enc.batch_decode_anchors(z, *cfg.anchors[:4]).cpu().numpy()
    assert np.array_equal(d0[0], OA.decode_anchors(np.zeros((cfg.num_anchors, 4), np.float32), ref[:4], [0.1, 0.1, 0.2, 0.2]))
    p = (np.random.RandomState(3).randn(2, cfg.num_anchors, 4) * 0.5).astype(np.float32)
    d_ref = OA.decode_anchors(p, ref[:4], [0.1, 0.1, 0.2, 0.2])
    d = cfg.enc.batch_decode_anchors(torch.from_numpy(p).to(dev), *cfg.anchors[:4]).cpu().numpy()
    assert np.allclose(d, d_ref, rtol=3e-7, atol=2e-4)               # expf last-ulp differences scaled by anchor size


@pytest.mark.parametrize("mining", [True, False])
def test_batched_encode_equals_the_per_image_calls_bit_for_bit(mining, dev):
    """danhip_encode_anchors_batched (one call, images side by side) against the per-image entry points the oracle tests above pin:
    ragged gt lists, an empty list (the [[0,0,1,1]] substitute of anchor_manipulator.py:286), many tiny faces (hard-face compensation
    with ties), and a 100-face image."""
    from dan_amd.train_sfd import AnchorConfig
    h = w = 320
    cfg = AnchorConfig(h, w, dev, match_threshold=0.4 if mining else 0.35, neg_threshold=0.4 if mining else 0.35)
    rs = np.random.RandomState(17)
    tiny = np.stack([np.array([y, x, y + s, x + s], np.float32) for y, x, s in zip(rs.randint(0, 300, 30), rs.randint(0, 300, 30), rs.randint(3, 14, 30))])
    gts = [_gt(12, h, w, 5), np.zeros((0, 4), np.float32), tiny, _gt(1, h, w, 6), _gt(100, h, w, 7), tiny[:3].repeat(2, axis=0)]
    gts = [torch.from_numpy(g).to(dev) for g in gts]
    T, L, S, M = cfg.enc.encode_anchors_batch(gts, *cfg.anchors, match_mining=mining)
    for b, g in enumerate(gts):
        t, l, s, m = cfg.enc.encode_anchors(g, *cfg.anchors, match_mining=mining)
        assert torch.equal(L[b], l) and torch.equal(S[b], s) and torch.equal(M[b], m) and torch.equal(T[b], t), b
    assert (L[2] == 1).sum().item() > 0


def test_batched_pa_encode_equals_the_per_image_calls(dev):
    from dan_amd.train_sfd import AnchorConfig
    h = w = 320
    cfg = AnchorConfig(h, w, dev)
    n0 = cfg.num_anchors_per_layer[0]
    sub = tuple(t[n0:].contiguous() for t in cfg.anchors)
    gts = [torch.from_numpy(_gt(n, h, w, 20 + n)).to(dev) for n in (7, 1, 33)] + [torch.zeros((0, 4), device=dev)]
    T, L, S, M = cfg.enc.encode_pa_anchors_batch(gts, *sub, 0.35, 0.35, match_mining=False, scale=2.)
    for b, g in enumerate(gts):
        t, l, s, m = cfg.enc.encode_pa_anchors(g, *sub, 0.35, 0.35, match_mining=False, scale=2.)
        assert torch.equal(L[b], l) and torch.equal(S[b], s) and torch.equal(M[b], m) and torch.equal(T[b], t), b


@pytest.mark.parametrize("scale", [2.0, 4.0])
@pytest.mark.parametrize("mining", [False, True])
@pytest.mark.parametrize("hw", [(320, 320), (640, 640)])
def test_encode_pa_anchors_vs_oracle(scale, mining, hw, dev):
    """AnchorEncoder.encode_pa_anchors (anchor_manipulator.py:328-387; PyramidBox head / body targets, train_pb.py:218,223): anchors of
    levels 1.. / 2.. shrunk by `scale` around their centre for matching, targets log(gt * scale / anchor).  HIP path (per-image entry
    AND the batched one, kernel branch anchors_exact.hip `scale != 1`) against oracle.anchors.encode_pa_anchors: labels, scores,
    matched boxes and centre offsets bit-exact; the two log() targets within 2 ulp of logf scaled by 1 / 0.2.  Cases: ragged lists,
    an empty list (the [[0,0,1,1]] substitute, :347), many tiny faces, one face, 100 faces."""
    from dan_amd.train_sfd import AnchorConfig
    h, w = hw
    ref = _cfg(h, w, 1.0)
    cfg = AnchorConfig(h, w, dev)
    first = sum(cfg.num_anchors_per_layer[:1 if scale == 2.0 else 2])           # head anchors: levels 1.., body: levels 2.. (train_pb.py:205-247)
    sub = tuple(t[first:].contiguous() for t in cfg.anchors)
    rsub = tuple(a[first:] for a in ref[:4])
    rinside = ref[4][first:]
    rs = np.random.RandomState(17)
    tiny = np.stack([np.array([y, x, y + s, x + s], np.float32) for y, x, s in zip(rs.randint(0, h - 20, 30), rs.randint(0, w - 20, 30), rs.randint(3, 14, 30))])
    gts = [_gt(12, h, w, 5), np.zeros((0, 4), np.float32), tiny, _gt(1, h, w, 6), _gt(100, h, w, 7)]
    mf = (lambda ov: OE.small_mining_match(ov, 0., 0.35, 0.35, 6, 0.3)) if mining else (lambda ov: OA.do_dual_max_match(ov, 0.35, 0.35))
    T, L, S, M = cfg.enc.encode_pa_anchors_batch([torch.from_numpy(g).to(dev) for g in gts], *sub, 0.35, 0.35, match_mining=mining, scale=scale)
    n_pos = 0
    for b, gt in enumerate(gts):
        t_ref, l_ref, s_ref, m_ref = OA.encode_pa_anchors(gt, rsub, rinside, [0.1, 0.1, 0.2, 0.2], mf, scale)
        t, l, s, m = cfg.enc.encode_pa_anchors(torch.from_numpy(gt).to(dev), *sub, 0.35, 0.35, match_mining=mining, scale=scale)
        for (tt, ll, ss, mm), what in (((t, l, s, m), "per-image"), ((T[b], L[b], S[b], M[b]), "batched")):
            assert np.array_equal(ll.cpu().numpy().astype(np.int64), l_ref), (b, what)
            assert np.array_equal(ss.cpu().numpy(), s_ref), (b, what)
            assert np.array_equal(mm.cpu().numpy(), m_ref), (b, what)
            tg = tt.cpu().numpy()
            assert np.array_equal(tg[:, :2], t_ref[:, :2]), (b, what)
            assert np.allclose(tg[:, 2:], t_ref[:, 2:], rtol=0, atol=4 * np.finfo(np.float32).eps * 5 * 4), (b, what)
        n_pos += int((l_ref == 1).sum())
    assert n_pos > 0


def test_face_scores_kernel_matches_the_oracle_softmax(dev):
    """danhip_face_scores: softmax(cls)[:, 1] of two-way logits and the easy-anchor mask (score > 0.03), against oracle.anchors.softmax_np
    (exp(x - max) / sum in fp32) — large and tiny logits included; the mask agrees wherever the score is not within 1e-6 of the threshold."""
    import numpy as np
    import torch
    from dan_amd import ops
    from oracle import anchors as OA
    g = torch.Generator().manual_seed(77)
    cls = torch.randn((3, 5000, 2), generator=g) * 6.0
    cls[0, :10] = torch.tensor([[80.0, -80.0], [-80.0, 80.0], [0.0, 0.0], [1e-8, -1e-8], [30.0, 29.999]]).repeat(2, 1)
    score, mask = ops.face_scores(cls.to(dev), 0.03)
    only = ops.face_scores(cls.to(dev))
    ref = OA.softmax_np(cls.numpy())[..., 1]
    got = score.cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-6
    assert torch.equal(only, score) and mask.dtype == torch.int32
    sure = np.abs(ref - 0.03) > 1e-6
    assert np.array_equal(mask.cpu().numpy()[sure], (ref > np.float32(0.03)).astype(np.int32)[sure])
