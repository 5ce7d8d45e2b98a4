"""North-star tolerance "eval_dan.py box outputs within 1e-4 of the reference on identical weights/inputs": the evaluation graphs on the fp32
inference path (csrc/f32_infer.hip; model.precision = "fp32") against the fp32 CPU oracle, END TO END — image -> logits -> decoded boxes
(+ dynamic anchor routing for DAN) — on three image sizes including ragged ones.

Tolerances (the oracle is itself "parity unpinned" against TensorFlow; what is pinned here is fp32 agreement of two implementations):
  logits   max|d| <= 1e-4 * max|ref|          (measured ~1e-6: accumulation order only)
  boxes    |d| <= 1e-4 * max(1, |ref|) px     for every anchor (S3FD, PyramidBox, stage-1 DAN boxes)
  DAN routed boxes: the same bound for every anchor whose routing decision agrees; the routing arg-max / thresholds are discrete, so a
  logit difference of 1e-6 may flip a cell — at most 0.1 % of the anchors may differ.
The bf16 build's logits sit at 4-6 % of scale on the same graphs (tests/test_models_gpu.py): that path cannot meet this bound."""
import numpy as np
import pytest
import torch

from oracle import anchors as OA
from oracle import extra_lib as OX
from oracle import nets as ON

pytestmark = pytest.mark.gpu
SIZES = [(96, 128), (160, 160), (203, 331)]
PS = [0.1, 0.1, 0.2, 0.2]


def _weights(forward, x, seed, deform=False):
    P = ON.Params(create=True, seed=seed)
    with torch.no_grad():
        forward(P, x)
    g = torch.Generator().manual_seed(99)
    for n in P.t:
        if n.endswith("/bias"):
            P.t[n] = 0.05 * torch.randn(P.t[n].shape, generator=g)
        if deform and n.endswith("deform_conv/conv2d/kernel"):
            P.t[n] = 0.02 * torch.randn(P.t[n].shape, generator=g)
        if deform and n.endswith("deform_conv/conv2d/bias"):
            P.t[n] = 0.6 * torch.randn(P.t[n].shape, generator=g)
    return P


def _close_logits(got, want, name):
    scale = want.abs().max().item()
    err = (got.cpu() - want).abs().max().item()
    assert err <= 1e-4 * scale, (name, err, scale)
    return err / scale


def _box_bad(got, ref):
    return np.abs(got - ref) > 1e-4 * np.maximum(1.0, np.abs(ref))


@pytest.mark.parametrize("h,w", SIZES)
@pytest.mark.parametrize("which", ["sfd", "pb"])
def test_single_stage_eval_boxes_fp32(which, h, w, dev):
    single_stage_case(which, h, w, dev, "fp32")


def single_stage_case(which, h, w, dev, precision):
    from dan_amd import synthetic
    from dan_amd.train_pb import PBModel
    from dan_amd.train_sfd import AnchorConfig, SFDModel
    imgs = synthetic.make_images(1, h, w, "cpu", seed=h + w)
    x = ON.preprocess_synthetic(imgs)
    ofwd = ON.sfd_forward if which == "sfd" else (lambda P, xx: ON.pb_forward(P, xx)["face"])
    P = _weights(ON.sfd_forward if which == "sfd" else ON.pb_forward, x, 7)
    with torch.no_grad():
        loc_r, cls_r = ofwd(ON.Params(P.t), x)
    model = (SFDModel if which == "sfd" else PBModel)(device=dev)
    model.vs.load_tf_named(P.t)
    model.precision = precision
    anchors = AnchorConfig(h, w, dev)
    with torch.no_grad():
        out = model.forward(imgs.to(dev))
        loc, cls = out if which == "sfd" else out["face"]
        boxes, scores = model.predict(imgs.to(dev), anchors)
    assert loc.dtype == torch.float32
    _close_logits(loc, loc_r, "loc")
    _close_logits(cls, cls_r, "cls")
    a4 = [t.cpu().numpy() for t in anchors.anchors[:4]]
    ref_b = OA.decode_anchors(loc_r[0].numpy(), a4, PS)
    ref_s = OA.softmax_np(cls_r[0].numpy())[:, 1]
    assert not _box_bad(boxes[0].cpu().numpy(), ref_b).any(), np.abs(boxes[0].cpu().numpy() - ref_b).max()
    assert np.allclose(scores[0].cpu().numpy(), ref_s, atol=1e-5)


@pytest.mark.parametrize("h,w", SIZES)
@pytest.mark.parametrize("deform", [False, True])
def test_dan_eval_boxes_fp32(deform, h, w, dev):
    """eval_dan.py:344-404 (the tensors fetched at :99): stage-1 boxes of levels 2.., routed stage-2 boxes of every level, scores."""
    dan_eval_case(deform, h, w, dev)


def dan_eval_case(deform, h, w, dev, logits16_tol=None, precision="fp32"):
    """One image through the DAN evaluation graph on the fp32 path against the oracle (logits, stage-1 boxes, routed stage-2 boxes, scores);
    logits16_tol: also compare the 16-bit path's four logit tensors at that fraction of the reference scale (tests/test_size_1024_gpu.py)."""
    from dan_amd import synthetic
    from dan_amd.train_dan import DANModel, dan_anchor_config
    imgs = synthetic.make_images(1, h, w, "cpu", seed=h + w + 1)
    x = ON.preprocess_synthetic(imgs)
    fwd = lambda P, xx: ON.dan_forward(P, xx, deform=deform)
    P = _weights(fwd, x[:, :64, :64], 9, deform)             # (variable shapes do not depend on the image size)
    with torch.no_grad():
        (l1r, c1r), (l2r, c2r) = fwd(ON.Params(P.t), x)
    model = DANModel(device=dev, deform=deform)
    model.vs.load_tf_named(P.t)
    anchors = dan_anchor_config(h, w, dev)
    if logits16_tol is not None:
        with torch.no_grad():
            (a1, b1), (a2, b2), _ = model.forward(imgs.to(dev))
        for got, want, name in ((a1, l1r, "stage1/loc"), (b1, c1r, "stage1/cls"), (a2, l2r, "stage2/loc"), (b2, c2r, "stage2/cls")):
            err = (got.float().cpu() - want).abs().max().item()
            assert err <= logits16_tol * want.abs().max().item(), (name, err, want.abs().max().item())
        del a1, b1, a2, b2
    model.precision = precision
    with torch.no_grad():
        (l1, c1), (l2, c2), sizes = model.forward(imgs.to(dev))
        boxes, scores = model.predict(imgs.to(dev), anchors)
    for got, want, name in ((l1, l1r, "stage1/loc"), (c1, c1r, "stage1/cls"), (l2, l2r, "stage2/loc"), (c2, c2r, "stage2/cls")):
        _close_logits(got, want, name)
    # ---- oracle assembly on the ORACLE's logits (numpy decode + C++ routing)
    a4 = [t.cpu().numpy() for t in anchors.anchors[:4]]
    loc1, cls1, loc2, cls2 = l1r[0].numpy(), c1r[0].numpy(), l2r[0].numpy(), c2r[0].numpy()
    s1, s2 = OA.softmax_np(cls1)[:, -1], OA.softmax_np(cls2)[:, -1]
    dec = OA.decode_anchors(loc1, a4, PS)
    outs_b, outs_s, off = [], [], 0
    for i, nl in enumerate(anchors.num_anchors_per_layer):
        sl = slice(off, off + nl)
        mo, do = OX.dynamic_anchor_routing(dec[sl], loc2[sl] / np.asarray([20., 20., 10., 10.], np.float32), s2[sl], (s1[sl] > 0.03).astype(np.int32),
                                           sizes[i][0], sizes[i][1], 1, [4, 8, 16, 32, 64, 128][i], h, w, False, 0.03, 0.0)
        outs_b.append(do)
        outs_s.append(s2[sl] * mo.astype(np.float32))
        off += nl
    first = sum(anchors.num_anchors_per_layer[:2])
    ref_b = np.concatenate([dec[first:]] + outs_b, 0)
    ref_s = np.concatenate([s1[first:]] + outs_s, 0)
    got_b, got_s = boxes[0].cpu().numpy(), scores[0].cpu().numpy()
    assert got_b.shape == ref_b.shape
    n1 = dec.shape[0] - first                                  # stage-1 part: no discrete decision
    assert not _box_bad(got_b[:n1], ref_b[:n1]).any()
    assert np.allclose(got_s[:n1], ref_s[:n1], atol=1e-5)
    bad = _box_bad(got_b[n1:], ref_b[n1:]).any(-1) | (np.abs(got_s[n1:] - ref_s[n1:]) > 1e-5)
    assert bad.mean() <= 1e-3, (int(bad.sum()), bad.size)


@pytest.mark.parametrize("N,H,W,Cin,Cout,k,stride,relu,resid", [
    (2, 96, 96, 64, 128, 3, 1, True, False),
    (1, 125, 167, 3, 96, 3, 1, True, False),         # Cin = 3 (scalar loads, one ragged 16-channel chunk), ragged M, ragged Cout
    (2, 100, 100, 64, 72, 3, 1, False, True),        # ragged Cout, residual after the activation
    (2, 150, 150, 32, 256, 1, 1, True, False),       # pointwise
    (4, 129, 127, 24, 128, 3, 2, True, False),       # stride 2, TF's asymmetric 'same' padding, C % 16 != 0
])
def test_fp32_conv_entry_point_vs_the_oracle(N, H, W, Cin, Cout, k, stride, relu, resid, dev):
    """danhip_conv2d_fwd_f32 (the evaluation graphs in fp32: "boxes within 1e-4") alone against oracle.tf_ops.conv2d_same in fp32 - 1e-5 of the
    output scale, the accumulation order being the only difference: ragged M / Cin / Cout, pointwise, stride 2, residual after the activation."""
    import ctypes
    from dan_amd import _lib, ops
    from oracle import tf_ops as T
    g = torch.Generator().manual_seed(N * 31 + Cin)
    x = torch.randn((N, H, W, Cin), generator=g)
    w = torch.randn((k, k, Cin, Cout), generator=g) / (k * k * Cin) ** 0.5
    b = torch.randn((Cout,), generator=g)
    want = T.conv2d_same(x, w, b, stride=stride, relu=relu)
    r = torch.randn(want.shape, generator=g) if resid else None
    if resid:
        want = want + r
    d = ops._desc(N, H, W, Cin, Cout, k, k, stride)
    xd, wd, bd = x.to(dev), w.to(dev), b.to(dev)
    rd = r.to(dev) if resid else None
    y = torch.full(want.shape, float("nan"), device=dev)
    _lib.call("danhip_conv2d_fwd_f32", ctypes.byref(d), _lib.ptr(xd), _lib.ptr(wd), _lib.ptr(bd), _lib.ptr(y), int(relu), _lib.ptr(rd), _lib.stream())
    torch.cuda.synchronize()
    got = y.cpu()
    assert torch.isfinite(got).all()
    assert (got - want).abs().max().item() <= 1e-5 * want.abs().max().item()
