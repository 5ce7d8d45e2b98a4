"""Parity hardening of round 6.

(a) The BENCHMARK's own shape — 16 images of 640 x 640 per GPU (BASELINE.json configs[1]) — had never been compared with anything: every
    full-size test ran one image.  Backward is linear in the loss's gradient, and no S3FD layer mixes images, so the parameter gradient
    of the 16-image batch equals the sum over i of the gradient image i produces ALONE (batch-1 kernels: the ones the full-size oracle
    tests pin) from its slice of the batch's loss gradient.  What this exercises: tiles that straddle images, the batch index of every
    kernel's addressing, the batch-dependent kernel / split-K choices, the pool-only training forward.
(b) A 20-step TRAJECTORY: every training-parity test was one step.  The bf16 HIP trainer and the oracle in bf16-storage emulation start
    from the same weights and take 20 Momentum steps on the same batch; the loss curves stay within 3 % and the parameters' movement agrees.
"""
import pytest
import torch

from oracle import nets as ON
from oracle import train as OT

pytestmark = pytest.mark.gpu


def test_batch16_gradient_at_640_is_the_sum_of_its_sixteen_images(dev):
    from dan_amd import ops, synthetic
    from dan_amd.train_sfd import AnchorConfig, SFDModel, SFDTrainer
    B, S = 16, 640
    imgs = synthetic.make_images(B, S, S, dev, seed=20180817)
    gts = synthetic.make_gt_boxes(B, S, S, seed=11, max_faces=12)
    model = SFDModel(device=dev, seed=3)
    anchors = AnchorConfig(S, S, dev)
    loc_t, cls_t, _ = anchors.encode_batch(gts)
    tr = SFDTrainer(model)                                   # flat gradient buffer: the backward kernels accumulate into it
    g = torch.Generator(device="cpu").manual_seed(5)
    for n, p in model.vs.named():                            # non-zero biases: every ReLU / pool decision away from the all-zero tie
        if n.endswith("/bias"):
            p.data.copy_((0.05 * torch.randn(p.shape, generator=g)).to(dev))
    ops.WEIGHT_EPOCH += 1
    # ---- the batch: forward, the loss's gradient with respect to the logits, backward
    tr.flat.zero_grad()
    loc, cls = model.forward(imgs)
    loc_d, cls_d = loc.detach().requires_grad_(True), cls.detach().requires_grad_(True)
    acc = ops.detection_loss(cls_d, loc_d, cls_t, loc_t, ratio=3.0, at_least_one=False, scale=1.0)
    acc.backward(torch.ones_like(acc))
    dloc, dcls = loc_d.grad.clone(), cls_d.grad.clone()
    assert dloc.abs().max().item() > 0 and dcls.abs().max().item() > 0
    torch.autograd.backward([loc, cls], [dloc, dcls])
    torch.cuda.synchronize()
    g16 = tr.flat.g.clone()
    del loc, cls
    # ---- image by image, each alone, from its slice of the SAME loss gradient, accumulated in the same buffer
    tr.flat.zero_grad()
    for i in range(B):
        li, ci = model.forward(imgs[i:i + 1].contiguous())
        torch.autograd.backward([li, ci], [dloc[i:i + 1].contiguous(), dcls[i:i + 1].contiguous()])
        del li, ci
    torch.cuda.synchronize()
    gsum = tr.flat.g
    assert torch.isfinite(g16).all() and torch.isfinite(gsum).all()
    bad = []
    for name, start, size in zip(tr.flat.names, tr.flat.starts, tr.flat.sizes):
        a, b = g16[start:start + size], gsum[start:start + size]
        scale = b.abs().max().item()
        if scale < 1e-12:
            continue
        # activations and their gradients are per-image quantities, identical bit patterns in both runs unless a layer changes its
        # kernel form with the batch size (split-K on the 20 x 20 ... 5 x 5 maps at one image: a data gradient re-rounded to bf16, 2^-9 per
        # element, carried through the layers below); the weight gradients differ by fp32 summation order.  Measured: every variable
        # <= 1e-2 except additional_layers/conv6_1 (2.0e-2); a dropped image or a mis-addressed tile is >= 6e-2 (1/16 of the sum)
        rel = (a - b).norm().item() / (b.norm().item() + 1e-30)
        if rel > 4e-2:
            bad.append((name, rel))
    assert not bad, bad[:8]
    cos = torch.nn.functional.cosine_similarity(g16, gsum, dim=0).item()
    assert cos > 0.9999, cos


def test_twenty_step_trajectory_tracks_the_oracle(dev):
    from dan_amd import synthetic
    from dan_amd.train_sfd import AnchorConfig, SFDModel, SFDTrainer
    B, S, STEPS = 2, 128, 20
    P = ON.Params(create=True, seed=4321)
    imgs = synthetic.make_images(B, S, S, "cpu", seed=17)
    x = ON.preprocess_synthetic(imgs)
    with torch.no_grad():
        ON.sfd_forward(P, x)
    g = torch.Generator().manual_seed(7)
    for n in P.t:
        if n.endswith("/bias"):
            P.t[n] = 0.05 * torch.randn(P.t[n].shape, generator=g)
    model = SFDModel(device=dev)
    model.vs.load_tf_named(P.t)
    anchors = AnchorConfig(S, S, dev)
    loc_t, cls_t, _ = anchors.encode_batch(synthetic.make_gt_boxes(B, S, S, seed=5, max_faces=6))
    lr = 1e-4                                                 # the reference's own first segment: 1e-3 x 0.1 for steps <= 1000 (train_sfd.py:98-109, 429-434)
    tr = SFDTrainer(model)
    w0 = {n: p.detach().clone().cpu() for n, p in model.vs.named()}
    # ---- oracle trajectory: bf16-storage emulation (weights / activations rounded as the HIP path stores them), fp32 master weights
    params = {n: v.clone() for n, v in P.t.items()}
    mom = {n: torch.zeros_like(v) for n, v in params.items()}
    xo = x.to(torch.bfloat16).float()
    lt, ct = loc_t.cpu(), cls_t.cpu().long()
    ref_losses = []
    for _ in range(STEPS):
        leaf = {n: v.clone().requires_grad_(True) for n, v in params.items()}
        loc, cls = ON.sfd_forward(ON.Params(leaf, emulate_bf16=True), xo)
        ce, ll, _ = OT.detection_loss(cls, loc, ct, lt)
        loss = ce + ll + OT.l2_regularizer(leaf)
        grads = dict(zip(leaf.keys(), torch.autograd.grad(loss, list(leaf.values()))))
        ref_losses.append((ce.item(), ll.item()))
        with torch.no_grad():
            OT.momentum_sgd_step(params, grads, mom, lr)
    # ---- HIP trajectory
    got_losses = []
    dimgs = imgs.to(dev)
    for _ in range(STEPS):
        tr.train_step(dimgs, loc_t, cls_t)
        ce, ll, _, _ = tr.losses()
        got_losses.append((ce, ll))
    curves = [(k, round(a[0], 4), round(b[0], 4), round(a[1], 4), round(b[1], 4)) for k, (a, b) in enumerate(zip(got_losses, ref_losses))]
    for k, ((ce, ll), (rce, rll)) in enumerate(zip(got_losses, ref_losses)):
        assert abs(ce - rce) <= 0.03 * abs(rce) + 1e-3, ("cross entropy", k, ce, rce, curves)
        assert abs(ll - rll) <= 0.03 * abs(rll) + 1e-3, ("localisation", k, ll, rll, curves)
    assert ref_losses[-1][0] + ref_losses[-1][1] < ref_losses[0][0] + ref_losses[0][1], "the oracle's loss did not go down: the run says nothing"
    # the parameters' MOVEMENT over the 20 steps: whole-model relative distance and direction
    num = den = dot = nrm = 0.0
    for n, p in model.vs.named():
        d_hip = p.detach().cpu() - w0[n]
        d_ref = params[n] - w0[n]
        num += (d_hip - d_ref).pow(2).sum().item()
        den += d_ref.pow(2).sum().item()
        dot += (d_hip * d_ref).sum().item()
        nrm += d_hip.pow(2).sum().item()
    assert den > 0 and num ** 0.5 <= 0.25 * den ** 0.5, (num ** 0.5, den ** 0.5)
    assert dot / (nrm ** 0.5 * den ** 0.5) >= 0.97


def test_restore_from_a_bundle_the_repo_did_not_write(dev, tmp_path):
    """(d) f2: a trainer restores variables, Momentum slots and global_step from a TF-V2 bundle produced by the independent encoder
    (tests/tf_bundle_encoder.py: two shards, restart interval 1, CRC per tensor) — not by dan_amd.utility.checkpoint's own writer — and
    then computes, bit for bit, the logits of the model the values came from and continues its step count."""
    import numpy as np
    from tf_bundle_encoder import write_bundle
    from dan_amd import synthetic
    from dan_amd.train_sfd import SFDModel, SFDTrainer
    from dan_amd.utility import checkpoint as C
    src = SFDModel(device=dev, seed=101)
    imgs = synthetic.make_images(1, 128, 128, dev, seed=9)
    with torch.no_grad():
        loc_a, cls_a = src.forward(imgs)
    scope = "sfd"
    g = np.random.default_rng(0)
    tensors, slots = {}, {}
    for n, p in src.vs.named():
        tensors[scope + "/" + n] = p.detach().cpu().numpy()
        slots[n] = (1e-3 * g.standard_normal(tuple(p.shape))).astype(np.float32)
        tensors[scope + "/" + n + "/Momentum"] = slots[n]
    tensors["global_step"] = np.asarray(4321, dtype=np.int64)
    prefix = str(tmp_path / "model.ckpt-4321")
    write_bundle(prefix, tensors)
    r = C.CheckpointReader(prefix)
    some = [n for n in tensors if n.endswith("fc6/conv2d/kernel")][0]
    assert np.array_equal(r.get_tensor(some, verify=True), tensors[some])              # the 18.9 MB tensor: CRC verified by the repo's reader
    dst = SFDModel(device=dev, seed=202)
    tr = SFDTrainer(dst)
    with torch.no_grad():
        loc_b0, _ = dst.forward(imgs)
    assert not torch.equal(loc_b0, loc_a)
    restored = tr.restore(prefix, scope)
    assert "global_step" in restored and tr.step_no == 4321
    with torch.no_grad():
        loc_b, cls_b = dst.forward(imgs)
    assert torch.equal(loc_b, loc_a) and torch.equal(cls_b, cls_a)
    mv = C._momentum_views(tr)
    for n in list(slots)[:5] + list(slots)[-5:]:
        assert np.array_equal(mv[n].detach().cpu().numpy(), slots[n]), n


@pytest.mark.parametrize("which,H,W", [("sfd", 96, 96), ("pb", 64, 64), ("dan", 64, 96), ("dan_deform", 64, 96)])
def test_free_running_decisions_flip_rarely_and_only_near_zero(which, H, W, dev, monkeypatch):
    """(c) The tight gradient tests IMPOSE the HIP forward's discrete decisions on the oracle, so a wrong decision is invisible there
    (VERDICT r5 weak 2).  Here nothing is imposed: the oracle (bf16-storage emulation) and the HIP path each take their own ReLU decisions on the
    same weights and image, and the test bounds how often they differ and where — per ReLU layer at most 1.5 % of the elements flip (measured,
    profiles/r6/decision_flips.jsonl: S3FD <= 0.20 %, PyramidBox / DAN <= 0.78 %, DAN-Deform <= 0.74 %; medians 0.05-0.14 %), and every flipped
    element is small in BOTH runs (|activation| <= 2 % of the layer's maximum; measured <= 0.8 %: a pre-activation within 16-bit rounding of zero,
    not a wrong decision).  DAN-Deform: 8 % (measured 5.3 % behind a deformable convolution) — the sampling cell floor(position) of an offset
    that was itself rounded to 16 bits is a discrete decision upstream of the ReLU, and a moved cell changes the pre-activation by more than
    rounding does.  With the flip rate bounded, what the imposed tests cover is the remaining 98 %+ of each layer.  DANHIP_TEST_REPORT_DIR:
    the per-graph worst layers are written there (profiles/r6/decision_flips.jsonl has the measured figures)."""
    import gradcheck as GC
    from dan_amd import ops
    model, flat, ofwd, P, imgs, x = GC.setup(which, H, W, 2, dev, torch.bfloat16)
    own = {}
    real_relu = ON._relu

    def recording_relu(Pp, y, scope):
        out = real_relu(Pp, y, scope)
        own[scope + "/kernel"] = out.detach()
        return out

    monkeypatch.setattr(ON, "_relu", recording_relu)
    with torch.no_grad():
        flat(ofwd(ON.Params({n: v.clone() for n, v in P.t.items()}, emulate_bf16=True), x))
    monkeypatch.setattr(ON, "_relu", real_relu)
    trace = {}
    ops.TRACE = {}
    try:
        model.forward(imgs.to(dev))      # (gradients tracked: the training graph — at inference DAN fuses a residual add into some ReLU convolutions'
        rec, ops.TRACE = ops.TRACE, None  #  epilogues, and what TRACE then holds for them is relu(conv) + x, not a ReLU output)
    finally:
        ops.TRACE = None
    GC.collect_trace(model.vs.named(), rec, trace)
    layers = [n for n in trace["relu"] if n in own and trace["relu"][n].shape == own[n].shape]
    assert len(layers) >= (13 if which == "sfd" else 30), (len(layers), len(trace["relu"]), len(own))
    worst = []
    for n in layers:
        a, b = trace["relu"][n], own[n]
        if a.min().item() < 0:                     # (what the HIP path recorded for this kernel is not a bare ReLU output: a fused residual)
            continue
        flip = (a > 0) != (b > 0)
        rate = flip.float().mean().item()
        scale = max(b.abs().max().item(), 1e-12)
        size = (torch.maximum(a.abs(), b.abs())[flip].max().item() / scale) if flip.any() else 0.0
        worst.append((rate, size, n))
    worst.sort(reverse=True)
    import json, os
    rep = os.environ.get("DANHIP_TEST_REPORT_DIR")
    if rep:
        os.makedirs(rep, exist_ok=True)
        with open(os.path.join(rep, "decision_flips.jsonl"), "a") as f:
            f.write(json.dumps({"graph": which, "relu_layers": len(worst), "max_flip_rate": worst[0][0], "median_flip_rate": sorted(w[0] for w in worst)[len(worst) // 2],
                                "max_flipped_activation_over_layer_max": max(w[1] for w in worst), "worst_layers": [[round(r, 5), round(z, 5), n] for r, z, n in worst[:3]]}) + "\n")
    assert len(worst) >= (13 if which == "sfd" else 30)
    assert worst[0][0] <= 0.015, worst[:5]
    assert max(w[1] for w in worst) <= (0.08 if which == "dan_deform" else 0.02), sorted(worst, key=lambda w: -w[1])[:5]
