"""fp16 build of libdanhip (DANHIP_DTYPE=fp16 -> libdanhip_f16.so, v_mfma_f32_16x16x32_f16): run by tests/test_fp16_gpu.py in a
child process because the activation dtype is a per-process property.  BASELINE.json configs[4] ("DAN-Deform ... fp16 + MFMA").
Tolerances are 4x tighter than the bf16 suite's (11 vs 8 significant bits)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
assert os.environ.get("DANHIP_DTYPE") == "fp16"

import torch

from dan_amd import _lib, ops, synthetic
from oracle import nets as ON
from oracle import tf_ops as T

assert _lib.ACT_DTYPE == torch.float16 and _lib.lib().danhip_act_dtype() == 2
T.EMULATE_DTYPE = torch.float16
dev = torch.device("cuda:0")
H = torch.float16


def conv_case(shape, seed):
    N, Hh, W, Cin, Cout, k, s = shape
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((N, Hh, W, Cin), generator=g).to(H)
    w = (torch.randn((k, k, Cin, Cout), generator=g) / (k * k * Cin) ** 0.5).to(H).float()
    b = torch.randn((Cout,), generator=g)
    xr, wr = x.float().requires_grad_(True), w.clone().requires_grad_(True)
    pre = T.conv2d_same(xr, wr, b, stride=s, relu=False)
    ref = torch.relu(pre)
    dy = torch.randn(ref.shape, generator=g).to(H)
    xd = x.to(dev).requires_grad_(True)
    wd = w.to(dev).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True)
    y = ops.conv2d(xd, wd, bd, stride=s, relu=True)
    y.backward(dy.to(dev))
    # the ReLU mask is taken from the device output: pre-activations within accumulation noise of zero may land on either side
    pre.backward(dy.float() * (y.detach().float().cpu() > 0))
    for name, got, want in (("y", y, ref), ("dx", xd.grad, xr.grad), ("dw", wd.grad, wr.grad)):
        want = want.detach()
        err = (got.detach().float().cpu() - want).abs().max().item()
        tol = 2.0 ** -9 * want.abs().max().item() + 2e-4
        assert err <= tol, (shape, name, err, tol)
    assert y.dtype == H and xd.grad.dtype == H


def main():
    n = 0
    # one shape per conv kernel family: halo 128-wide, halo 16x16, 64->64 register-resident, 3->64 first layer, flat-M 1x1,
    # stride 2, thin head (halo) and ragged Cout
    for i, shp in enumerate([(2, 32, 64, 128, 128, 3, 1), (1, 32, 32, 256, 256, 3, 1), (2, 16, 64, 64, 64, 3, 1), (2, 17, 45, 8, 64, 3, 1),
                             (1, 12, 12, 512, 256, 1, 1), (2, 20, 20, 128, 256, 3, 2), (1, 32, 32, 64, 8, 3, 1), (1, 9, 7, 72, 24, 3, 1)]):
        conv_case(shp, 10 + i)
        n += 1

    # HBM-bound layers
    g = torch.Generator().manual_seed(5)
    x = torch.randn((2, 15, 21, 64), generator=g).to(H)
    xr = x.float().requires_grad_(True)
    gamma = (torch.rand(64, generator=g) * 10).requires_grad_(True)
    ref = T.max_pool_2x2_same(T.round_bf16(T.l2_normalize(xr, gamma), True, True))   # the intermediate is stored in fp16
    dy = torch.randn(ref.shape, generator=g).to(H)
    ref.backward(dy.float())
    xd = x.to(dev).requires_grad_(True)
    gd = gamma.detach().to(dev).requires_grad_(True)
    y = ops.max_pool_2x2(ops.l2_normalize(xd, gd))
    y.backward(dy.to(dev))
    assert (y.float().cpu() - ref.detach()).abs().max().item() <= 2.0 ** -9 * ref.abs().max().item()
    assert (xd.grad.float().cpu() - xr.grad).abs().max().item() <= 2.0 ** -8 * xr.grad.abs().max().item() + 1e-3
    n += 1

    # S3FD forward against the oracle in fp16-storage emulation
    from dan_amd.train_sfd import AnchorConfig, SFDModel
    P = ON.Params(create=True, seed=1234)
    imgs = synthetic.make_images(1, 96, 128, "cpu", seed=7)
    xin = ON.preprocess_synthetic(imgs)
    with torch.no_grad():
        ON.sfd_forward(P, xin)
        loc_ref, cls_ref = ON.sfd_forward(ON.Params(P.t, emulate_bf16=True), xin.to(H).float())
    model = SFDModel(device=dev)
    model.vs.load_tf_named(P.t)
    with torch.no_grad():
        loc, cls = model.forward(imgs.to(dev))
    for got, want, name in ((loc, loc_ref, "loc"), (cls, cls_ref, "cls")):
        err = (got.cpu() - want).abs().max().item()
        assert err <= 0.01 * want.abs().max().item(), (name, err, want.abs().max().item())
    n += 1

    # DAN-Deform (configs[4]) trains: dynamic loss scale (starts at 1024) in the backward pass, finite losses, parameters move
    from dan_amd.train_dan import DANModel, DANTrainer, dan_anchor_config, encode_batch_dan
    model = DANModel(device=dev, deform=True)
    anchors = dan_anchor_config(128, 128, dev)
    tr = DANTrainer(model, anchors, world=1)
    assert tr.loss_scale == 1.0 and tr.ls_state.tolist() == [1024.0, 0.0, 1000.0, 0.0]      # the scale lives on the device
    imgs = synthetic.make_images(2, 128, 128, dev, seed=3)
    gts = synthetic.make_gt_boxes(2, 128, 128, seed=4, max_faces=5)
    targets = encode_batch_dan(anchors, gts)
    w0 = tr.flat.w.clone()
    totals = []
    for _ in range(3):
        tr.train_step(imgs, *targets)
        lv = tr.loss_values()
        totals.append(lv["total"])
        assert all(map(lambda v: v == v and abs(v) < 1e4, [lv["total"], lv["l2"]])), lv
    assert torch.isfinite(tr.flat.w).all() and torch.isfinite(tr.flat.g).all()
    assert (tr.flat.w - w0).abs().max().item() > 0
    assert tr.flat.g.abs().max().item() > 1.0          # gradients are still in loss-scale units inside the flat buffer
    assert tr.ls_state.tolist() == [1024.0, 3.0, 1000.0, 0.0]                               # three clean steps, none skipped
    # ... and they ARE the loss-scaled gradients: the same first step with loss_scale = 1 gives 1/1024 of them (fp16 rounding and the
    # underflow the scale exists to avoid aside)
    norms = []
    for s in (1024.0, 1.0):
        m2 = DANModel(device=dev, deform=True)
        t2 = DANTrainer(m2, anchors, world=1, loss_scale=s, dynamic_loss_scale=False)
        t2.train_step(imgs, *targets)
        norms.append(t2.flat.g.norm().item())
    assert 0.8 < norms[0] / (1024.0 * norms[1]) < 1.25, norms
    n += 1
    # graph-level gradient parity (tests/gradcheck.py: fixed upstream gradient on every head output, per variable) in the fp16 build: the same
    # graph code and kernels as the bf16 suite at 3 more bits of storage precision, so the per-variable noise floor drops from 0.25 to
    # 0.12 for >= 97 % of the variables (the oracle's own fp32-vs-fp16-emulated gradients differ by 0.03 median / 0.11 max on these graphs)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import gradcheck as GC
    for which, hh, ww in (("pb", 64, 64), ("dan", 64, 96), ("dan_deform", 64, 96)):
        model, flat, ofwd, P, imgs, x = GC.setup(which, hh, ww, 2, dev, H)
        want, Gs, outs_ref = GC.oracle_grads(ofwd, flat, P, x)
        Gs = [g_ * 0.25 for g_ in Gs]                   # keep the largest activation gradients well inside fp16's range
        want = {k: (v * 0.25 if v is not None else None) for k, v in want.items()}
        got, outs = GC.hip_grads(model, flat, imgs, Gs, dev)
        for o, r in zip(outs, outs_ref):
            assert (o - r).abs().max().item() <= 0.015 * r.abs().max().item(), which
        bad, checked = GC.compare(got, want, 0.12)
        worst, _ = GC.compare(got, want, 0.25)
        # (the 1x1 .. 2x3 maps of pyramid levels 3-5 at this input size hold a handful of ReLU decisions each: a few of their variables sit
        # above the typical floor)
        assert checked > 250 and not worst and len(bad) <= 0.03 * checked, (which, checked, worst[:10], bad[:10])
        n += 1
    # BASELINE.json configs[4] at its input size: DAN-Deform on ONE 1024 x 1024 image in the fp16 build — the four 16-bit logit tensors against the
    # fp32 oracle (fp16 storage: 2 % of scale), the fp32 path's logits / stage-1 boxes at 1e-4 and the routed stage-2 boxes
    import test_eval_f32_gpu as TE
    T.EMULATE_DTYPE = torch.bfloat16          # (the helper compares against the plain fp32 oracle: leave the emulation switch as it found it)
    TE.dan_eval_case(True, 1024, 1024, dev, logits16_tol=0.02)
    T.EMULATE_DTYPE = torch.float16
    n += 1
    # ... and ONE full training step of DAN-Deform at 1024 x 1024 (configs[4]'s size and build): loss terms + every variable's gradient with the
    # forward decisions imposed on the oracle (tests/gradcheck.py::train_step_case)
    checked = GC.train_step_case("dan_deform", 1024, 1024, dev, H, grad_tol=0.05)
    assert checked > 100
    n += 1
    print("FP16-OK", n, "groups; DAN-Deform losses", ["%.4f" % t for t in totals])


if __name__ == "__main__":
    main()
