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
        # kernel form with the batch size (bf16 re-rounding of a data gradient: 2^-9); the weight gradients differ by fp32 summation order
        rel = (a - b).norm().item() / (b.norm().item() + 1e-30)
        if rel > 2e-2:
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
    lr = 1e-3                                                 # (x 0.1 of the schedule's first segment would move nothing in 20 steps)
    tr = SFDTrainer(model, base_lr=lr, lr_factors=(1.0, 1.0, 0.1, 0.01))
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
    for k, ((ce, ll), (rce, rll)) in enumerate(zip(got_losses, ref_losses)):
        assert abs(ce - rce) <= 0.03 * abs(rce) + 1e-3, ("cross entropy", k, ce, rce)
        assert abs(ll - rll) <= 0.03 * abs(rll) + 1e-3, ("localisation", k, ll, rll)
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
