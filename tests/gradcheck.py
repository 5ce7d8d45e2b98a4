"""Shared body of the graph-level gradient parity checks (tests/test_grad_parity_gpu.py for the bf16 build, tests/fp16/cases.py for
the fp16 build): PyramidBox / DAN / DAN-Deform, HIP path against the CPU oracle graph in 16-bit-storage emulation, identical weights
and inputs, per variable.

The upstream gradient is a FIXED random tensor on every head output (loss = sum(out * G)): no hard-negative mining, so the comparison
isolates the backward GRAPH (GradSlot hand-off, torch.cat / residual glue, stop_gradient) from threshold effects of the loss."""
import torch

from oracle import nets as ON


def make_weights(forward, x, seed, deform=False):
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


def setup(which, H, W, B, dev, act_dtype, seed=11):
    """-> (model, trainer-less forward fn returning the flat list of head outputs, oracle forward fn, P, imgs, x)"""
    from dan_amd import synthetic
    deform = which == "dan_deform"
    imgs = synthetic.make_images(B, H, W, "cpu", seed=3)
    x = ON.preprocess_synthetic(imgs)
    if which == "sfd":
        from dan_amd.train_sfd import SFDModel
        ofwd = ON.sfd_forward
        flat = lambda o: [o[0], o[1]]
        P = make_weights(ofwd, x, seed)
        model = SFDModel(device=dev)
    elif which == "pb":
        from dan_amd.train_pb import PBModel
        ofwd = ON.pb_forward
        flat = lambda o: [o[k][j] for k in ("face", "head", "body") for j in (0, 1)]
        P = make_weights(ofwd, x, seed)
        model = PBModel(device=dev)
    else:
        from dan_amd.train_dan import DANModel
        ofwd = lambda P_, xx: ON.dan_forward(P_, xx, deform=deform)
        flat = lambda o: [o[0][0], o[0][1], o[1][0], o[1][1]]
        P = make_weights(ofwd, x, seed, deform)
        model = DANModel(device=dev, deform=deform)
    model.vs.load_tf_named(P.t)
    return model, flat, ofwd, P, imgs, x.to(act_dtype).float()


def oracle_grads(ofwd, flat, P, x, Gs=None, seed=5, impose=None):
    params = {n: v.clone().requires_grad_(True) for n, v in P.t.items()}
    PO = ON.Params(params, emulate_bf16=True)
    PO.impose = impose
    outs = flat(ofwd(PO, x))
    if Gs is None:
        gen = torch.Generator().manual_seed(seed)
        Gs = [torch.randn(o.shape, generator=gen) for o in outs]
    sum((o * G).sum() for o, G in zip(outs, Gs)).backward()
    return {n: p.grad for n, p in params.items()}, Gs, [o.detach() for o in outs]


def hip_grads(model, flat, imgs, Gs, dev, trace=None):
    """Gradients of sum(out * G) w.r.t. every variable through the HIP path (plain autograd leaves: no flat buffer).
    trace: a dict that receives the forward pass's discrete decisions in the form oracle.nets.Params.impose takes."""
    from dan_amd import ops
    named = model.vs.named()
    for _, p in named:
        p.grad = None
    ops.TRACE = {} if trace is not None else None
    try:
        outs = flat(model.forward(imgs.to(dev)))
        rec, ops.TRACE = ops.TRACE, None
    finally:
        ops.TRACE = None
    if trace is not None:
        name_of = {id(p): n for n, p in named}
        trace["pools"] = [t.float().cpu() for t in rec.pop("pools", [])]
        trace["maxout"] = {name_of[k]: v.float().cpu() for k, v in rec.pop("maxout", {}).items() if k in name_of}
        trace["offsets"] = {name_of[k]: v.float().cpu() for k, v in rec.pop("offsets", {}).items() if k in name_of}
        trace["relu"] = {name_of[k]: v.float().cpu() for k, v in rec.items() if k in name_of}
    torch.autograd.backward(outs, [G.to(dev) for G in Gs])
    torch.cuda.synchronize()
    return {n: (p.grad.detach().cpu() if p.grad is not None else None) for n, p in named}, [o.detach().cpu() for o in outs]


def compare(got, want, tol, skip=1e-6):
    """-> (list of (name, rel) beyond tol, number compared).  Variables without an oracle gradient must have none / zero on the HIP path."""
    bad, checked = [], 0
    for name, w in want.items():
        g = got[name]
        if w is None or w.abs().max().item() == 0.0:
            assert g is None or g.abs().max().item() == 0.0, (name, "oracle has no gradient, the HIP path has one")
            continue
        assert g is not None, (name, "no gradient on the HIP path")
        if w.abs().max().item() <= skip:
            continue
        rel = (g.reshape(-1) - w.reshape(-1)).norm().item() / (w.norm().item() + 1e-12)
        checked += 1
        if rel > tol:
            bad.append((name, round(rel, 4)))
    return bad, checked
