"""The four graphs at the BASELINE spatial size (640 x 640, one image — BASELINE.json configs[0] is "S3FD forward, one 640x640 image"),
END TO END against the CPU oracle: the 16-bit path's logits at the tolerance tests/test_models_gpu.py uses at 64-200 px (accumulation
depth does not grow with the image: the bound holds unchanged), and the fp32 inference path's decoded boxes at the north-star 1e-4."""
import numpy as np
import pytest
import torch

from oracle import anchors as OA
from oracle import nets as ON

pytestmark = pytest.mark.gpu
S = 640


def _weights(forward, x, seed):
    P = ON.Params(create=True, seed=seed)
    with torch.no_grad():
        forward(P, x)
    g = torch.Generator().manual_seed(99)
    for n in P.t:
        if n.endswith("/bias"):
            P.t[n] = 0.05 * torch.randn(P.t[n].shape, generator=g)
    return P


@pytest.mark.parametrize("which", ["sfd", "pb", "dan", "dan_deform"])
def test_graph_at_640_fp32_boxes_and_16bit_logits(which, dev):
    from dan_amd import synthetic
    from dan_amd.train_dan import DANModel, dan_anchor_config
    from dan_amd.train_pb import PBModel
    from dan_amd.train_sfd import AnchorConfig, SFDModel
    imgs = synthetic.make_images(1, S, S, "cpu", seed=640)
    x = ON.preprocess_synthetic(imgs)
    if which == "sfd":
        ofwd = ON.sfd_forward
    elif which == "pb":
        ofwd = lambda P, xx: ON.pb_forward(P, xx)["face"]
    else:
        ofwd = lambda P, xx: ON.dan_forward(P, xx, deform=(which == "dan_deform"))[0]      # stage-1 logits (no discrete routing decision)
    full = {"sfd": ON.sfd_forward, "pb": ON.pb_forward}.get(which, lambda P, xx: ON.dan_forward(P, xx, deform=(which == "dan_deform")))
    P = _weights(full, x, 21)
    with torch.no_grad():
        loc_r, cls_r = ofwd(ON.Params(P.t), x)
    model = {"sfd": SFDModel, "pb": PBModel}.get(which, lambda device: DANModel(device=device, deform=(which == "dan_deform")))(device=dev)
    model.vs.load_tf_named(P.t)
    anchors = (dan_anchor_config if which.startswith("dan") else AnchorConfig)(S, S, dev)

    def logits():
        with torch.no_grad():
            out = model.forward(imgs.to(dev))
        if which == "sfd":
            return out
        if which == "pb":
            return out["face"]
        return out[0]

    # ---- 16-bit path
    loc, cls = logits()
    for got, want, name in ((loc, loc_r, "loc"), (cls, cls_r, "cls")):
        err = (got.float().cpu() - want).abs().max().item()
        assert err <= 0.06 * want.abs().max().item(), (which, name, err, want.abs().max().item())
    # ---- fp32 path: logits 1e-4 of scale, decoded boxes 1e-4 px-relative
    model.precision = "fp32"
    loc32, cls32 = logits()
    model.precision = "act"
    assert loc32.dtype == torch.float32
    for got, want, name in ((loc32, loc_r, "loc"), (cls32, cls_r, "cls")):
        err = (got.cpu() - want).abs().max().item()
        assert err <= 1e-4 * want.abs().max().item(), (which, name, err)
    a4 = [t.cpu().numpy() for t in anchors.anchors[:4]]
    ref_b = OA.decode_anchors(loc_r[0].numpy(), a4, [0.1, 0.1, 0.2, 0.2])
    got_b = OA.decode_anchors(loc32[0].cpu().numpy(), a4, [0.1, 0.1, 0.2, 0.2])
    assert got_b.shape[0] == anchors.num_anchors
    assert (np.abs(got_b - ref_b) <= 1e-4 * np.maximum(1.0, np.abs(ref_b))).all()


def test_s3fd_training_step_at_640(dev):
    """BASELINE.json configs[1]'s shape (640 x 640; one image so that the CPU oracle finishes in seconds): loss terms and the gradient of
    EVERY variable against the oracle in bf16-storage emulation, at the per-tensor bound tests/test_sfd_gpu.py uses at 128 x 128."""
    import test_sfd_gpu as TS
    from oracle import train as OT
    from dan_amd.train_sfd import SFDTrainer
    P, model, anchors, imgs, x, loc_t, cls_t = TS._setup(dev, B=1, H=S, W=S)
    tr = SFDTrainer(model, world=1)
    params = {n: v.clone().requires_grad_(True) for n, v in P.t.items()}
    PO = ON.Params(params, emulate_bf16=True)
    loc_ref, cls_ref = ON.sfd_forward(PO, x.to(torch.bfloat16).float())
    ce, locl, _ = OT.detection_loss(cls_ref, loc_ref, cls_t.cpu().long(), loc_t.cpu())
    (ce + locl).backward()
    tr.train_step(imgs.to(dev), loc_t, cls_t)
    g_ce, g_loc, g_l2, _ = tr.losses()
    assert abs(g_ce - ce.item()) <= 0.03 * abs(ce.item()) + 1e-3, (g_ce, ce.item())
    assert abs(g_loc - locl.item()) <= 0.03 * abs(locl.item()) + 1e-3, (g_loc, locl.item())
    bad = []
    for name, prm in dict(model.vs.named()).items():
        got = prm.grad.detach().reshape(-1).cpu()
        want = params[name].grad.reshape(-1)
        rel = (got - want).norm().item() / (want.norm().item() + 1e-8)
        if rel > 0.08 and want.abs().max().item() > 1e-6:
            bad.append((name, rel))
    assert not bad, bad[:8]


@pytest.mark.parametrize("which,size", [("pb", 640), ("dan", 1024)])
def test_training_step_at_baseline_size_with_decisions_imposed(which, size, dev):
    """the FULL training step - every loss term and every variable's gradient - at the input size BASELINE.json quotes
    for the graph (PyramidBox configs[2]: 640 x 640; DAN configs[3]: 1024 x 1024; one image so the CPU oracle finishes in a minute), with the
    HIP forward's discrete decisions imposed on the oracle (tests/gradcheck.py::train_step_case): loss terms 3 %, gradients 0.05 per
    variable.  DAN-Deform at 1024 x 1024 runs in the fp16 build (tests/fp16/cases.py), the build configs[4] names."""
    import gradcheck as GC
    checked = GC.train_step_case(which, size, size, dev, torch.bfloat16)
    assert checked > 100
