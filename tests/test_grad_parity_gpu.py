"""Graph-level gradient parity of the PyramidBox / DAN / DAN-Deform training steps (HIP path, bf16 storage) against the CPU oracle
graphs run in bf16-storage emulation with identical weights, inputs and targets — per variable, like tests/test_sfd_gpu.py does for
S3FD.  This is where a dropped or duplicated contribution in the GradSlot hand-off (ops.py), the torch.cat / residual-add glue or the
stop_gradient of get_features_stage2 (net/danet.py:934) would show.

Tolerances (written per test): loss terms 3 % (hard-negative mining runs on each side's own logits, a bf16-level difference can move an
anchor across the mining threshold); per-variable gradient: relative L2 error <= 0.10 (0.15 for DAN-Deform: bilinear-sampling
derivatives amplify the bf16 rounding of the offsets), variables whose oracle gradient is below 1e-6 in max-norm are skipped."""
import pytest
import torch

from oracle import nets as ON
from oracle import train as OT

pytestmark = pytest.mark.gpu


def _weights(forward, x, seed, deform=False):
    P = ON.Params(create=True, seed=seed)
    with torch.no_grad():
        forward(P, x)
    g = torch.Generator().manual_seed(99)
    for n in P.t:
        if n.endswith("/bias"):
            P.t[n] = 0.05 * torch.randn(P.t[n].shape, generator=g)
    if deform:                                    # offsets are zero-initialised (custom_op.py:132): exercise the gather path too
        for n in P.t:
            if n.endswith("deform_conv/conv2d/kernel"):
                P.t[n] = 0.02 * torch.randn(P.t[n].shape, generator=g)
            if n.endswith("deform_conv/conv2d/bias"):
                P.t[n] = 0.6 * torch.randn(P.t[n].shape, generator=g)
    return P


def _compare(model, params, tol, skip=1e-6):
    bad, checked = [], 0
    for name, prm in model.vs.named():
        want = params[name].grad
        got = prm.grad.detach().reshape(-1).cpu()
        if want is None:
            assert got.abs().max().item() == 0.0, (name, "oracle has no gradient, HIP path has one")
            continue
        want = want.reshape(-1)
        if want.abs().max().item() <= skip:
            continue
        rel = (got - want).norm().item() / (want.norm().item() + 1e-12)
        checked += 1
        if rel > tol:
            bad.append((name, round(rel, 4)))
    return bad, checked


def _close(a, b, rel):
    return abs(a - b) <= rel * abs(b) + 1e-3


def test_pyramidbox_gradients_match_the_oracle(dev):
    from dan_amd import synthetic
    from dan_amd.train_pb import PBAnchorTargets, PBModel, PBTrainer
    H = W = 64
    B = 2
    imgs = synthetic.make_images(B, H, W, "cpu", seed=3)
    x = ON.preprocess_synthetic(imgs)
    P = _weights(ON.pb_forward, x, 11)
    model = PBModel(device=dev)
    model.vs.load_tf_named(P.t)
    tr = PBTrainer(model, world=1)
    tg = PBAnchorTargets(H, W, dev)
    targets = tg.encode_batch(synthetic.make_gt_boxes(B, H, W, seed=2, max_faces=3))
    # ---- oracle: total = face + 0.66 head + 0.33 body (train_pb.py:454-463), bf16-storage emulation
    params = {n: v.clone().requires_grad_(True) for n, v in P.t.items()}
    out = ON.pb_forward(ON.Params(params, emulate_bf16=True), x.to(torch.bfloat16).float())
    total, ref = 0.0, {}
    for k, wgt in (("face", 1.0), ("head", 0.66), ("body", 0.33)):
        loc, cls = out[k]
        ce, ll, _ = OT.detection_loss(cls, loc, targets[k][1].cpu().long(), targets[k][0].cpu())
        ref[k] = (ce.item(), ll.item())
        total = total + wgt * (ce + ll)
    total.backward()
    tr.train_step(imgs.to(dev), targets)
    vals = tr.loss_values()
    for k in ("face", "head", "body"):
        assert _close(vals[k][0], ref[k][0], 0.03) and _close(vals[k][1], ref[k][1], 0.03), (k, vals[k], ref[k])
    bad, checked = _compare(model, params, 0.10)
    assert checked > 250 and not bad, (checked, bad[:10])


@pytest.mark.parametrize("deform", [False, True])
def test_dan_gradients_match_the_oracle(deform, dev):
    from dan_amd import synthetic
    from dan_amd.train_dan import DANModel, DANTrainer, dan_anchor_config, encode_batch_dan
    H, W, B = 64, 96, 2
    imgs = synthetic.make_images(B, H, W, "cpu", seed=5)
    x = ON.preprocess_synthetic(imgs)
    fwd = lambda P, xx: ON.dan_forward(P, xx, deform=deform)
    P = _weights(fwd, x, 21, deform)
    model = DANModel(device=dev, deform=deform)
    model.vs.load_tf_named(P.t)
    anchors = dan_anchor_config(H, W, dev)
    tr = DANTrainer(model, anchors, world=1)
    loc_t, cls_t, mgt = encode_batch_dan(anchors, synthetic.make_gt_boxes(B, H, W, seed=5, max_faces=3))
    tr.train_step(imgs.to(dev), loc_t, cls_t, mgt)
    vals = tr.loss_values()
    fm, fl = tr.last_routing               # stage-2 targets: the routing of the HIP path's decoded stage-1 boxes (bit-exact kernels, test_routing_gpu.py)
    # ---- oracle: stage-1 loss + stage-2 loss on the routed targets (train_dan.py:470-489), at_least_one mining (:302)
    params = {n: v.clone().requires_grad_(True) for n, v in P.t.items()}
    (l1, c1), (l2, c2) = fwd(ON.Params(params, emulate_bf16=True), x.to(torch.bfloat16).float())
    ce1, ll1, _ = OT.detection_loss(c1, l1, cls_t.cpu().long(), loc_t.cpu(), at_least_one=True)
    ce2, ll2, _ = OT.detection_loss(c2, l2, fm.cpu().long(), fl.cpu(), at_least_one=True)
    (ce1 + ll1 + ce2 + ll2).backward()
    assert _close(vals["stage1"][0], ce1.item(), 0.03) and _close(vals["stage1"][1], ll1.item(), 0.03), (vals["stage1"], ce1.item(), ll1.item())
    assert _close(vals["stage2"][0], ce2.item(), 0.03) and _close(vals["stage2"][1], ll2.item(), 0.03), (vals["stage2"], ce2.item(), ll2.item())
    bad, checked = _compare(model, params, 0.15 if deform else 0.10)
    assert checked > 200 and not bad, (checked, bad[:10])


@pytest.mark.parametrize("deform", [False, True])
def test_stage2_loss_sends_no_gradient_into_stage1(deform, dev):
    """get_features_stage2 consumes tf.stop_gradient(stage-1 features) (net/danet.py:934, net/danet_deform.py:311): backward of the
    stage-2 loss ALONE must leave every variable that only feeds stage 1 (context blocks, lfpn_stage1, predict_face) at exactly zero,
    while the shared trunk (backbone, lfpn) and the stage-2 variables receive a gradient — on the HIP path and in the oracle."""
    from dan_amd import synthetic
    from dan_amd.train_dan import DANModel, DANTrainer, dan_anchor_config, encode_batch_dan
    H, W, B = 64, 96, 2
    imgs = synthetic.make_images(B, H, W, "cpu", seed=6)
    x = ON.preprocess_synthetic(imgs)
    fwd = lambda P, xx: ON.dan_forward(P, xx, deform=deform)
    P = _weights(fwd, x, 22, deform)
    model = DANModel(device=dev, deform=deform)
    model.vs.load_tf_named(P.t)
    anchors = dan_anchor_config(H, W, dev)
    tr = DANTrainer(model, anchors, world=1)
    loc_t, cls_t, mgt = encode_batch_dan(anchors, synthetic.make_gt_boxes(B, H, W, seed=7, max_faces=3))
    tr.flat.zero_grad()
    terms = tr.loss_terms(imgs.to(dev), loc_t, cls_t, mgt)
    acc2 = terms[1][2]
    torch.autograd.backward([acc2], [torch.ones_like(acc2)])
    torch.cuda.synchronize()
    fm, fl = tr.last_routing
    params = {n: v.clone().requires_grad_(True) for n, v in P.t.items()}
    (_, _), (l2, c2) = fwd(ON.Params(params, emulate_bf16=True), x.to(torch.bfloat16).float())
    ce2, ll2, _ = OT.detection_loss(c2, l2, fm.cpu().long(), fl.cpu(), at_least_one=True)
    (ce2 + ll2).backward()
    stage1_only = ("prediction_modules_stage1/", "lfpn_stage1/", "predict_face/")
    n_zero = n_live = 0
    for name, prm in model.vs.named():
        g = prm.grad.detach()
        og = params[name].grad
        if name.startswith(stage1_only):
            assert g.abs().max().item() == 0.0, (name, g.abs().max().item())
            assert og is None or og.abs().max().item() == 0.0, name
            n_zero += 1
        elif og is not None and og.abs().max().item() > 1e-6:
            assert g.abs().max().item() > 0.0, name
            n_live += 1
    assert n_zero > 40 and n_live > 100, (n_zero, n_live)
    bad, checked = _compare(model, params, 0.15 if deform else 0.10)
    assert not bad, bad[:10]
