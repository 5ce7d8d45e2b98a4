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
        collect_trace(named, rec, trace)
    torch.autograd.backward(outs, [G.to(dev) for G in Gs])
    torch.cuda.synchronize()
    return {n: (p.grad.detach().cpu() if p.grad is not None else None) for n, p in named}, [o.detach().cpu() for o in outs]


def collect_trace(named, rec, trace):
    """ops.TRACE of one forward pass -> the dict oracle.nets.Params.impose takes (+ "loss_sel": the hard-negative selections, call order)."""
    name_of = {id(p): n for n, p in named}
    trace["pools"] = [t.float().cpu() for t in rec.pop("pools", [])]
    trace["maxout"] = {name_of[k]: v.float().cpu() for k, v in rec.pop("maxout", {}).items() if k in name_of}
    trace["offsets"] = {name_of[k]: v.float().cpu() for k, v in rec.pop("offsets", {}).items() if k in name_of}
    trace["loss_sel"] = [t.cpu().bool() for t in rec.pop("loss_sel", [])]
    trace["relu"] = {name_of[k]: v.float().cpu() for k, v in rec.items() if k in name_of}
    return trace


def imposed_loss(cls_pred, loc_pred, cls_targets, loc_targets, sel, negative_ratio=3.0):
    """oracle.train.detection_loss (train_sfd.py:386-417) with the hard-negative SELECTION given instead of mined: the anchors that enter
    the cross-entropy are `sel`; positives (cls_targets > 0) carry the smooth-L1 term.  -> (ce, loc)."""
    import torch.nn.functional as F
    from oracle import train as OT
    pos = cls_targets > 0
    labels = torch.clamp(cls_targets[sel], 0, 2).to(torch.int64)
    ce = F.cross_entropy(cls_pred[sel], labels, reduction="mean") * (negative_ratio + 1.0)
    loc = OT.modified_smooth_l1(loc_pred[pos], loc_targets[pos]).sum(-1).mean()
    return ce, loc


def train_step_case(which, H, W, dev, act_dtype, grad_tol=0.05, loss_tol=0.03, seed=11):
    """ONE full training step (forward, every loss term, backward into the flat gradient buffer) of `which` on one H x W image through the
    product's trainer, against the oracle graph in 16-bit-storage emulation with the HIP forward's discrete decisions imposed: ReLU signs,
    2x2 arg-max positions, max-out choices, sampling offsets (DAN-Deform), the hard-negative selection of every loss term and - DAN - the
    stage-2 targets routed from the HIP path's stage-1 boxes (bit-exact kernels, tests/test_routing_gpu.py).  What remains is accumulation
    order and 16-bit rounding: loss terms within `loss_tol`, every variable's gradient within `grad_tol` relative L2.
    (reference: train_pb.py:400-504, train_dan.py:386-524.)  -> number of variables compared."""
    from dan_amd import ops, synthetic
    from oracle import train as OT
    model, flat, ofwd, P, imgs, x = setup(which, H, W, 1, dev, act_dtype, seed=seed)
    gts = synthetic.make_gt_boxes(1, H, W, seed=2, max_faces=6)
    named = model.vs.named()
    if which == "pb":
        from dan_amd.train_pb import PBAnchorTargets, PBTrainer
        tr = PBTrainer(model, world=1)
        targets = PBAnchorTargets(H, W, dev).encode_batch(gts)
        step_args = (imgs.to(dev), targets)
    elif which == "sfd":
        from dan_amd.train_sfd import AnchorConfig, SFDTrainer
        tr = SFDTrainer(model, world=1)
        loc_t, cls_t, _ = AnchorConfig(H, W, dev).encode_batch(gts)
        step_args = (imgs.to(dev), loc_t, cls_t)
    else:
        from dan_amd.train_dan import DANTrainer, dan_anchor_config, encode_batch_dan
        anchors = dan_anchor_config(H, W, dev)
        tr = DANTrainer(model, anchors, world=1)
        loc_t, cls_t, mgt = encode_batch_dan(anchors, gts)
        step_args = (imgs.to(dev), loc_t, cls_t, mgt)
    ops.TRACE = {}
    try:
        tr.train_step(*step_args)
        rec, ops.TRACE = ops.TRACE, None
    finally:
        ops.TRACE = None
    torch.cuda.synchronize()
    trace = collect_trace(named, rec, {})
    vals = tr.loss_values()
    scale = float(tr.ls_state[0].item()) if getattr(tr, "ls_state", None) is not None else float(getattr(tr, "loss_scale", 1.0))
    # ---- oracle
    params = {n: v.clone().requires_grad_(True) for n, v in P.t.items()}
    PO = ON.Params(params, emulate_bf16=True)
    PO.impose = trace
    out = ofwd(PO, x)
    sels = trace["loss_sel"]
    ref = {}
    if which == "pb":
        total = 0.0
        for i, (k, wgt) in enumerate((("face", 1.0), ("head", 0.66), ("body", 0.33))):
            ce, ll = imposed_loss(out[k][1], out[k][0], targets[k][1].cpu().long(), targets[k][0].cpu(), sels[i])
            ref[k] = (ce.item(), ll.item())
            total = total + wgt * (ce + ll)
    elif which == "sfd":
        ce, ll = imposed_loss(out[1], out[0], cls_t.cpu().long(), loc_t.cpu(), sels[0])
        ref["face"] = (ce.item(), ll.item())
        total = ce + ll
    else:
        fm, fl = tr.last_routing
        (l1, c1), (l2, c2) = out
        ce1, ll1 = imposed_loss(c1, l1, cls_t.cpu().long(), loc_t.cpu(), sels[0])
        ce2, ll2 = imposed_loss(c2, l2, fm.cpu().long(), fl.cpu(), sels[1])
        ref["stage1"], ref["stage2"] = (ce1.item(), ll1.item()), (ce2.item(), ll2.item())
        total = ce1 + ll1 + ce2 + ll2
    total.backward()
    for k, (ce, ll) in ref.items():
        key = k if k in vals else [kk for kk in vals if kk not in ("l2", "total")][0]
        assert abs(vals[key][0] - ce) <= loss_tol * abs(ce) + 1e-3 and abs(vals[key][1] - ll) <= loss_tol * abs(ll) + 1e-3, (which, k, vals[key], (ce, ll))
    got = {n: (p.grad.detach().float().cpu() / scale if p.grad is not None else None) for n, p in named}
    want = {n: p.grad for n, p in params.items()}
    bad, checked = compare(got, want, grad_tol)
    assert not bad, (which, H, W, checked, bad[:12])
    return checked


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
