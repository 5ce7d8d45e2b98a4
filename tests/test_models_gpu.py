"""Forward parity of the PyramidBox / DAN / DAN-Deform graphs (HIP path, bf16 activations) against the CPU oracle graphs
with identical weights and inputs, plus the eval_dan.py output assembly (decode + dynamic anchor routing).
Tolerance: 4 % of each output's scale (20-50 bf16 layers, fp32 accumulate; the oracle runs in bf16-storage emulation)."""
import numpy as np
import pytest
import torch

from oracle import anchors as OA
from oracle import extra_lib as OX
from oracle import nets as ON

pytestmark = pytest.mark.gpu


def _weights(forward, x, seed):
    P = ON.Params(create=True, seed=seed)
    with torch.no_grad():
        forward(P, x)
    g = torch.Generator().manual_seed(99)
    for n in P.t:
        if n.endswith("/bias"):
            P.t[n] = 0.05 * torch.randn(P.t[n].shape, generator=g)
    return P


def _check(got, want, name, rel=0.04):
    scale = want.abs().max().item()
    err = (got.cpu() - want).abs().max().item()
    assert err <= rel * scale, (name, err, scale)


def test_pyramidbox_forward_parity(dev):
    from dan_amd import synthetic
    from dan_amd.train_pb import PBModel
    imgs = synthetic.make_images(1, 64, 64, "cpu", seed=3)
    x = ON.preprocess_synthetic(imgs)
    P = _weights(ON.pb_forward, x, 11)
    with torch.no_grad():
        ref = ON.pb_forward(ON.Params(P.t, emulate_bf16=True), x.to(torch.bfloat16).float())
    model = PBModel(device=dev)
    model.vs.load_tf_named(P.t)
    with torch.no_grad():
        out = model.forward(imgs.to(dev))
    assert set(n for n, _ in model.vs.named()) == set(P.t.keys())
    for k in ("face", "head", "body"):
        assert out[k][0].shape == ref[k][0].shape and out[k][1].shape == ref[k][1].shape
        _check(out[k][0], ref[k][0], k + "/loc")
        _check(out[k][1], ref[k][1], k + "/cls")


@pytest.mark.parametrize("deform", [False, True])
def test_dan_forward_parity(deform, dev):
    from dan_amd import synthetic
    from dan_amd.train_dan import DANModel
    imgs = synthetic.make_images(1, 64, 96, "cpu", seed=5)
    x = ON.preprocess_synthetic(imgs)
    fwd = lambda P, xx: ON.dan_forward(P, xx, deform=deform)
    P = _weights(fwd, x, 21)
    if deform:                                    # offsets are zero-initialised (custom_op.py:132): exercise the gather path too
        g = torch.Generator().manual_seed(5)
        for n in P.t:
            if n.endswith("deform_conv/conv2d/kernel"):
                P.t[n] = 0.05 * torch.randn(P.t[n].shape, generator=g)
            if n.endswith("deform_conv/conv2d/bias"):
                P.t[n] = 0.8 * torch.randn(P.t[n].shape, generator=g)
    with torch.no_grad():
        (l1r, c1r), (l2r, c2r) = fwd(ON.Params(P.t, emulate_bf16=True), x.to(torch.bfloat16).float())
    model = DANModel(device=dev, deform=deform)
    model.vs.load_tf_named(P.t)
    with torch.no_grad():
        (l1, c1), (l2, c2), sizes = model.forward(imgs.to(dev))
    assert set(n for n, _ in model.vs.named()) == set(P.t.keys())
    assert sizes == [(16, 24), (8, 12), (4, 6), (2, 3), (1, 2), (1, 1)]
    rel = 0.06 if deform else 0.04
    _check(l1, l1r, "stage1/loc", rel)
    _check(c1, c1r, "stage1/cls", rel)
    _check(l2, l2r, "stage2/loc", rel)
    _check(c2, c2r, "stage2/cls", rel)


def test_dan_eval_output_assembly(dev):
    """eval_dan.py:373-404 on given head outputs: the decode + per-level routing + concat of the HIP path equals the oracle's
    (numpy decode + C++ routing) on the same logits."""
    from dan_amd import synthetic
    from dan_amd.train_dan import DANModel, dan_anchor_config
    H, W = 64, 96
    model = DANModel(device=dev)
    anchors = dan_anchor_config(H, W, dev)
    imgs = synthetic.make_images(2, H, W, dev, seed=8)
    g = torch.Generator().manual_seed(4)
    A = anchors.num_anchors
    fake = ((0.3 * torch.randn((2, A, 4), generator=g)).to(dev), (2.0 * torch.randn((2, A, 2), generator=g)).to(dev))
    fake2 = ((2.0 * torch.randn((2, A, 4), generator=g)).to(dev), (2.0 * torch.randn((2, A, 2), generator=g)).to(dev))
    sizes = anchors.shapes
    model.forward = lambda im: (fake, fake2, sizes)
    boxes, scores = model.predict(imgs, anchors)
    a4 = [t.cpu().numpy() for t in anchors.anchors[:4]]
    for b in range(2):
        loc1, cls1 = fake[0][b].cpu().numpy(), fake[1][b].cpu().numpy()
        loc2, cls2 = fake2[0][b].cpu().numpy(), fake2[1][b].cpu().numpy()
        s1 = OA.softmax_np(cls1)[:, -1]
        s2 = OA.softmax_np(cls2)[:, -1]
        dec = OA.decode_anchors(loc1, a4, [0.1, 0.1, 0.2, 0.2])
        outs_b, outs_s = [], []
        off = 0
        for i, nl in enumerate(anchors.num_anchors_per_layer):
            sl = slice(off, off + nl)
            mo, do = OX.dynamic_anchor_routing(dec[sl], loc2[sl] / np.asarray([20., 20., 10., 10.], np.float32), s2[sl], (s1[sl] > 0.03).astype(np.int32),
                                               sizes[i][0], sizes[i][1], 1, [4, 8, 16, 32, 64, 128][i], H, W, False, 0.03, 0.0)
            outs_b.append(do)
            outs_s.append(s2[sl] * mo.astype(np.float32))
            off += nl
        first = sum(anchors.num_anchors_per_layer[:2])
        ref_b = np.concatenate([dec[first:]] + outs_b, 0)
        ref_s = np.concatenate([s1[first:]] + outs_s, 0)
        got_b, got_s = boxes[b].cpu().numpy(), scores[b].cpu().numpy()
        assert got_b.shape == ref_b.shape
        # the easy-mask threshold (score > 0.03) and the routing arg-max are discrete: compare where the two score vectors agree
        assert np.allclose(got_s, ref_s, atol=2e-6)
        assert np.allclose(got_b, ref_b, rtol=1e-4, atol=1e-3), np.abs(got_b - ref_b).max()


@pytest.mark.parametrize("h,w", [(75, 133), (203, 331)])
def test_dan_forward_parity_ragged_sizes(h, w, dev):
    """eval-time image sizes are arbitrary (eval_dan.py:303 placeholder (None, None, 3)): SAME pools / stride-2 convs round
    up, the LFPN resize targets the lateral's size, and every conv tile is ragged at the right / bottom edge."""
    from dan_amd import synthetic
    from dan_amd.train_dan import DANModel
    from dan_amd.train_sfd import layer_shapes
    imgs = synthetic.make_images(1, h, w, "cpu", seed=h)
    x = ON.preprocess_synthetic(imgs)
    fwd = lambda P, xx: ON.dan_forward(P, xx, deform=False)
    P = _weights(fwd, x, 33)
    with torch.no_grad():
        (l1r, c1r), (l2r, c2r) = fwd(ON.Params(P.t, emulate_bf16=True), x.to(torch.bfloat16).float())
    model = DANModel(device=dev)
    model.vs.load_tf_named(P.t)
    with torch.no_grad():
        (l1, c1), (l2, c2), sizes = model.forward(imgs.to(dev))
    assert sizes == layer_shapes(h, w)
    for got, want, name in ((l1, l1r, "stage1/loc"), (c1, c1r, "stage1/cls"), (l2, l2r, "stage2/loc"), (c2, c2r, "stage2/cls")):
        assert got.shape == want.shape
        _check(got, want, name, 0.04)


def test_unused_variants_bi_lfpn_and_conv_only_heads(dev):
    """SURVEY §8f row 4, the variants that need no new kernel: build_bi_lfpn (net/danet.py:191-249) and the *_conv_only deformable
    heads (net/danet_deform.py:328-366) — forward parity against the oracle graph on identical weights."""
    from dan_amd import synthetic
    from dan_amd.net import danet_deform, sfd_net
    from dan_amd.net.variables import VariableStore
    imgs = synthetic.make_images(1, 64, 96, "cpu", seed=9)
    x = ON.preprocess_synthetic(imgs)

    def fwd(P, xx):
        feats = ON.get_featmaps(P, xx)
        bi = ON.build_bi_lfpn(P, feats, name="bi_lfpn")
        s1 = ON.features_conv_only(P, bi)
        s2 = ON.features_conv_only(P, bi, stage1=s1)
        return bi, s1, s2
    P = _weights(fwd, x, 41)
    g = torch.Generator().manual_seed(6)
    for n in P.t:                                  # non-zero offsets so the gather path is exercised
        if "_conv" in n and n.endswith("/conv2d/kernel") and "predict_stage" in n:
            P.t[n] = 0.004 * torch.randn(P.t[n].shape, generator=g)     # sub-pixel .. ~1 px offsets: their bf16 storage stays a small effect
    with torch.no_grad():
        bi_r, s1_r, s2_r = fwd(ON.Params(P.t, emulate_bf16=True), x.to(torch.bfloat16).float())
    vs = VariableStore(device=dev)
    b = danet_deform.VGG16Backbone("channels_last", variables=vs)
    with torch.no_grad():
        xin = sfd_net.prepare_input(imgs.to(dev))
        def run():
            feats = b.get_featmaps(xin, training=False)
            bi = b.build_bi_lfpn(feats, name="bi_lfpn")
            s1 = b.get_features_stage1_conv_only(bi)
            s2 = b.get_features_stage2_conv_only(s1, bi)
            return bi, s1, s2
        run()                                      # creates the variables
        vs.load_tf_named(P.t)
        bi, s1, s2 = run()
    assert set(n for n, _ in vs.named()) == set(P.t.keys())
    for name, got, want in (("bi_lfpn", bi, bi_r), ("stage1_conv_only", s1, s1_r), ("stage2_conv_only", s2, s2_r)):
        assert len(got) == len(want) == 6
        for i, (a, r) in enumerate(zip(got, want)):
            assert a.shape == r.shape
            # the conv_only heads sample un-normalised LFPN outputs (no ReLU before them): 2304-term sums of bf16-rounded samples
            # that largely cancel, so the error is judged against a looser fraction of the (small) output scale
            _check(a.float(), r, "%s[%d]" % (name, i), 0.06 if name == "bi_lfpn" else 0.12)


def test_unused_variant_reverse_lfpn(dev):
    """build_reverse_lfpn (net/danet.py:382-412; stride-2 3x3 convs on the 160 / 80 / 40 maps): forward parity against
    the oracle graph on identical weights, and gradients of a fixed upstream gradient against the oracle's (bf16-storage emulation)."""
    from dan_amd import synthetic
    from dan_amd.net import danet, sfd_net
    from dan_amd.net.variables import VariableStore
    imgs = synthetic.make_images(2, 96, 128, "cpu", seed=19)
    x = ON.preprocess_synthetic(imgs)

    def fwd(P, xx):
        return ON.build_reverse_lfpn(P, ON.get_featmaps(P, xx), skip_last=3)
    P = _weights(fwd, x, 43)
    params = {n: v.clone().requires_grad_(True) for n, v in P.t.items()}
    ref = fwd(ON.Params(params, emulate_bf16=True), x.to(torch.bfloat16).float())
    g = torch.Generator().manual_seed(2)
    Gs = [torch.randn(r.shape, generator=g) for r in ref[1:4]]
    sum((r * G).sum() for r, G in zip(ref[1:4], Gs)).backward()
    vs = VariableStore(device=dev)
    b = danet.VGG16Backbone("channels_last", variables=vs)
    xin = sfd_net.prepare_input(imgs.to(dev))
    with torch.no_grad():
        b.build_reverse_lfpn(b.get_featmaps(xin, training=False))        # creates the variables
    vs.load_tf_named(P.t)
    assert set(n for n, _ in vs.named()) == set(P.t.keys())
    out = b.build_reverse_lfpn(b.get_featmaps(xin, training=True))
    assert len(out) == len(ref) == 6 and [tuple(o.shape) for o in out] == [tuple(r.shape) for r in ref]
    for i, (a, r) in enumerate(zip(out, ref)):
        _check(a.float().detach(), r.detach(), "reverse_lfpn[%d]" % i, 0.04)
    torch.autograd.backward(list(out[1:4]), [G.to(dev).to(out[1].dtype) for G in Gs])
    bad = []
    for n, p in vs.named():
        want = params[n].grad
        if want is None or want.abs().max().item() < 1e-6:
            continue
        rel = (p.grad.detach().cpu().reshape(-1) - want.reshape(-1)).norm().item() / (want.norm().item() + 1e-12)
        if rel > 0.2:
            bad.append((n, round(rel, 3)))
    assert not bad, bad[:8]
