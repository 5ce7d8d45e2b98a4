"""S3FD end-to-end parity (forward, loss, gradients, one optimiser step) of the HIP path against the CPU oracle graph
with identical weights and inputs.  bf16 activations: tolerances are relative to each tensor's scale."""
import numpy as np
import pytest
import torch

from oracle import nets as ON
from oracle import train as OT

pytestmark = pytest.mark.gpu


def _setup(dev, B=2, H=128, W=128):
    from dan_amd import synthetic
    from dan_amd.train_sfd import AnchorConfig, SFDModel
    P = ON.Params(create=True, seed=1234)
    imgs = synthetic.make_images(B, H, W, "cpu", seed=7)
    x = ON.preprocess_synthetic(imgs)
    with torch.no_grad():
        ON.sfd_forward(P, x)           # creates the variables
    # give biases / l2-norm scales non-trivial values so their gradients are exercised
    g = torch.Generator().manual_seed(99)
    for n in P.t:
        if n.endswith("/bias"):
            P.t[n] = 0.05 * torch.randn(P.t[n].shape, generator=g)
    model = SFDModel(device=dev)
    model.vs.load_tf_named(P.t)
    anchors = AnchorConfig(H, W, dev)
    gts = synthetic.make_gt_boxes(B, H, W, seed=3, max_faces=6)
    loc_t, cls_t, _ = anchors.encode_batch(gts)
    return P, model, anchors, imgs, x, loc_t, cls_t


def test_sfd_forward_parity(dev):
    P, model, anchors, imgs, x, loc_t, cls_t = _setup(dev)
    with torch.no_grad():
        loc_ref, cls_ref = ON.sfd_forward(P, x)
        loc, cls = model.forward(imgs.to(dev))
    for got, want, name in ((loc, loc_ref, "loc"), (cls, cls_ref, "cls")):
        scale = want.abs().max().item()
        err = (got.cpu() - want).abs().max().item()
        assert err <= 0.04 * scale, (name, err, scale)


def test_sfd_train_step_parity(dev):
    from dan_amd.train_sfd import SFDTrainer
    P, model, anchors, imgs, x, loc_t, cls_t = _setup(dev)
    tr = SFDTrainer(model, world=1)
    # ---- oracle loss + gradients (fp32)
    params = {n: v.clone().requires_grad_(True) for n, v in P.t.items()}
    PO = ON.Params(params, emulate_bf16=True)          # same storage precision as the HIP path (bf16 weights/activations)
    loc_ref, cls_ref = ON.sfd_forward(PO, x.to(torch.bfloat16).float())
    ce, locl, _ = OT.detection_loss(cls_ref, loc_ref, cls_t.cpu().long(), loc_t.cpu())
    (ce + locl).backward()
    w_before = {n: p.detach().clone().cpu() for n, p in model.vs.named()}
    acc = tr.train_step(imgs.to(dev), loc_t, cls_t)
    g_ce, g_loc, g_l2, g_total = tr.losses()
    assert abs(g_ce - ce.item()) <= 0.03 * abs(ce.item()) + 1e-3, (g_ce, ce.item())
    assert abs(g_loc - locl.item()) <= 0.03 * abs(locl.item()) + 1e-3, (g_loc, locl.item())
    l2_ref = OT.l2_regularizer(P.t).item()
    assert abs(g_l2 - l2_ref) <= 1e-4 * l2_ref
    # gradients: global relative error per tensor (bf16 network, fp32 oracle)
    bad = []
    named = dict(model.vs.named())                    # loc_i | cls_i live as strided views of one fused block: go through the parameters
    for name, prm in named.items():
        got = prm.grad.detach().reshape(-1).cpu()
        want = params[name].grad.reshape(-1)
        denom = want.norm().item() + 1e-8
        rel = (got - want).norm().item() / denom
        if rel > 0.08 and want.abs().max().item() > 1e-6:
            bad.append((name, rel))
    assert not bad, bad[:8]
    # optimiser step: w' = w - lr * (g*mult + wd*w)  (first step: v = g)
    lr = 1e-4   # step 0 of the schedule: 1e-3 * 0.1
    order = [n for n, _ in model.vs.named()]
    for name in order[:6] + order[-4:]:
        g = named[name].grad.detach().reshape(-1).cpu()
        w0 = w_before[name].reshape(-1)
        mult = 2.0 if "/bias" in name else 1.0
        wd = 0.0 if "/bias" in name else (0.2 * 5e-4 if "l2_norm_layer" in name else 5e-4)
        want = w0 - lr * (g + wd * w0) * mult
        got = named[name].detach().reshape(-1).cpu()
        assert torch.allclose(got, want, rtol=1e-5, atol=1e-7), name
