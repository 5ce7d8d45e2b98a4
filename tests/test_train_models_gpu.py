"""Training-step checks of the PyramidBox / DAN / DAN-Deform harnesses (train_pb.py / train_dan.py equivalents): every loss
term of the HIP step equals the oracle's loss (hard-negative mining + CE*(ratio+1) + smooth-L1) evaluated on the HIP
path's own head outputs and targets (so the comparison isolates mining / loss / routing-target plumbing), gradients reach
every variable, and the optimizer moves the weights.  Tolerance: 1e-3 relative (fp32 reductions in a different order)."""
import pytest
import torch

from oracle import train as OT

pytestmark = pytest.mark.gpu


def _close(a, b, rel=2e-3):
    return abs(a - b) <= rel * max(abs(b), 1e-3) + 1e-5


def test_pyramidbox_train_step(dev):
    from dan_amd import synthetic
    from dan_amd.train_pb import PBAnchorTargets, PBModel, PBTrainer
    H = W = 64
    model = PBModel(device=dev, seed=3)
    tr = PBTrainer(model, world=1)
    tg = PBAnchorTargets(H, W, dev)
    imgs = synthetic.make_images(2, H, W, dev, seed=1)
    targets = tg.encode_batch(synthetic.make_gt_boxes(2, H, W, seed=2, max_faces=3))
    assert targets["face"][0].shape == (2, 342, 4) and targets["head"][0].shape == (2, 86, 4) and targets["body"][0].shape == (2, 22, 4)
    with torch.no_grad():
        out = model.forward(imgs)
    w0 = tr.flat.w.clone()
    tr.train_step(imgs, targets)
    vals = tr.loss_values()
    for k in ("face", "head", "body"):
        loc, cls = out[k]
        ce, ll, _ = OT.detection_loss(cls.cpu(), loc.cpu(), targets[k][1].cpu().long(), targets[k][0].cpu())
        assert _close(vals[k][0], ce.item()) and _close(vals[k][1], ll.item()), (k, vals[k], ce.item(), ll.item())
    total = sum(w * (vals[k][0] + vals[k][1]) for k, w in (("face", 1.0), ("head", 0.66), ("body", 0.33))) + vals["l2"]
    assert _close(vals["total"], total)
    assert torch.isfinite(tr.flat.g).all() and (tr.flat.g != 0).float().mean().item() > 0.5
    assert (tr.flat.w != w0).any()


@pytest.mark.parametrize("deform", [False, True])
def test_dan_train_step(deform, dev):
    from dan_amd import synthetic
    from dan_amd.train_dan import DANModel, DANTrainer, dan_anchor_config, encode_batch_dan
    H, W = 64, 96
    model = DANModel(device=dev, seed=4, deform=deform)
    anchors = dan_anchor_config(H, W, dev)
    tr = DANTrainer(model, anchors, world=1)
    imgs = synthetic.make_images(2, H, W, dev, seed=1)
    loc_t, cls_t, mgt = encode_batch_dan(anchors, synthetic.make_gt_boxes(2, H, W, seed=5, max_faces=3))
    with torch.no_grad():
        (loc1, cls1), (loc2, cls2), _ = model.forward(imgs)
    tr.train_step(imgs, loc_t, cls_t, mgt)
    vals = tr.loss_values()
    ce1, ll1, _ = OT.detection_loss(cls1.cpu(), loc1.cpu(), cls_t.cpu().long(), loc_t.cpu(), at_least_one=True)
    assert _close(vals["stage1"][0], ce1.item()) and _close(vals["stage1"][1], ll1.item()), (vals["stage1"], ce1.item(), ll1.item())
    fm, fl = tr.last_routing
    assert set(fm.unique().tolist()) <= {-1, 0, 1} and (fm == 1).sum().item() > 0
    ce2, ll2, _ = OT.detection_loss(cls2.cpu(), loc2.cpu(), fm.cpu().long(), fl.cpu(), at_least_one=True)
    assert _close(vals["stage2"][0], ce2.item()) and _close(vals["stage2"][1], ll2.item()), (vals["stage2"], ce2.item(), ll2.item())
    assert torch.isfinite(tr.flat.g).all()
    # stop_gradient(stage-1 features) in get_features_stage2 (danet.py:934): the deform offset conv of stage 2 still trains
    names = [n for n, _ in model.vs.named()]
    assert any("prediction_modules_stage2" in n for n in names)
    if deform:
        gi = {n: q.grad for n, q in model.vs.named()}
        assert gi["prediction_modules_stage1/predict_stage1_0/deform_conv/kernel"].abs().sum().item() > 0
        assert gi["prediction_modules_stage1/predict_stage1_0/deform_conv/conv2d/bias"].abs().sum().item() > 0


def test_graph_captured_step_equals_eager_steps(dev):
    """DetectorTrainer.enable_graph: the whole step (forward, backward, fused SGD, batched weight repack) replayed as one hipGraph
    must walk the same trajectory as the eager launches (up to fp32-atomics order in the weight gradients)."""
    from dan_amd import synthetic
    from dan_amd.train_sfd import AnchorConfig, SFDModel, SFDTrainer
    imgs = synthetic.make_images(2, 128, 128, dev, seed=11)
    gts = synthetic.make_gt_boxes(2, 128, 128, seed=12, max_faces=5)
    anchors = AnchorConfig(128, 128, dev)
    loc_t, cls_t, _ = anchors.encode_batch(gts)
    runs = []
    for graph in (False, True):
        tr = SFDTrainer(SFDModel(device=dev, seed=5), world=1)
        if graph:
            tr.enable_graph(imgs, loc_t, cls_t, warmup=2)
        else:
            for _ in range(2):
                tr.train_step(imgs, loc_t, cls_t)
        losses = []
        for _ in range(3):
            tr.train_step(imgs, loc_t, cls_t)
            losses.append(tr.loss_values()["total"])
        assert tr.step_no == 5
        runs.append((tr.flat.w.clone(), losses))
    (w0, l0), (w1, l1) = runs
    assert all(abs(a - b) <= 2e-3 * abs(a) for a, b in zip(l0, l1)), (l0, l1)
    assert (w0 - w1).abs().max().item() <= 2e-3 * w0.abs().max().item()
    assert l0[-1] < l0[0]                                  # and it trains


def test_weight_gradient_stream_does_not_change_the_trajectory(dev, monkeypatch):
    """ops.wgrad_overlap_begin/join: weight gradients issued on a second stream (default for single-process training) against the same
    steps with everything on one stream — gradients after a step and weights after three, up to fp32-atomics order."""
    from dan_amd import ops, synthetic
    from dan_amd.train_sfd import AnchorConfig, SFDModel, SFDTrainer
    imgs = synthetic.make_images(3, 160, 128, dev, seed=31)
    gts = synthetic.make_gt_boxes(3, 160, 128, seed=32, max_faces=6)
    anchors = AnchorConfig(160, 128, dev)
    loc_t, cls_t, _ = anchors.encode_batch(gts)
    runs = []
    for on in (True, False):
        monkeypatch.setattr(ops, "WGRAD_STREAM", on)
        tr = SFDTrainer(SFDModel(device=dev, seed=6), world=1)
        tr.train_step(imgs, loc_t, cls_t)
        torch.cuda.synchronize()
        assert not ops.context().wgrad["on"] and not ops.context().wgrad["keep"]          # joined, nothing kept alive
        g1 = tr.flat.g.clone()
        for _ in range(2):
            tr.train_step(imgs, loc_t, cls_t)
        runs.append((g1, tr.flat.w.clone()))
    (ga, wa), (gb, wb) = runs
    assert (ga - gb).abs().max().item() <= 1e-3 * ga.abs().max().item()
    assert (wa - wb).abs().max().item() <= 1e-3 * wa.abs().max().item()
    assert ga.abs().max().item() > 0


def test_loss_scale_reaches_the_backward_and_cancels_in_the_optimizer(dev):
    """DetectorTrainer(loss_scale=s): the backward pass runs on s * loss (fp16 build: s = 1024) and danhip_sgd_momentum_flat divides
    the weight gradients by s again — the flat gradient buffer is s times the unscaled one and the trajectory is unchanged
    (s a power of two: exact in bf16 up to atomics order)."""
    from dan_amd import synthetic
    from dan_amd.train_sfd import AnchorConfig, SFDModel, SFDTrainer
    imgs = synthetic.make_images(2, 128, 128, dev, seed=41)
    gts = synthetic.make_gt_boxes(2, 128, 128, seed=42, max_faces=5)
    anchors = AnchorConfig(128, 128, dev)
    loc_t, cls_t, _ = anchors.encode_batch(gts)
    runs = []
    for s in (1.0, 1024.0):
        tr = SFDTrainer(SFDModel(device=dev, seed=7), world=1, loss_scale=s)
        tr.train_step(imgs, loc_t, cls_t)
        g1 = tr.flat.g.clone()
        tr.train_step(imgs, loc_t, cls_t)
        runs.append((g1, tr.flat.w.clone()))
    (g_a, w_a), (g_b, w_b) = runs
    assert (g_b / 1024.0 - g_a).abs().max().item() <= 2e-3 * g_a.abs().max().item()
    assert (w_a - w_b).abs().max().item() <= 1e-3 * w_a.abs().max().item()


def test_dynamic_loss_scale_entry_skips_nonfinite_steps_and_follows_the_growth_rule(dev):
    """danhip_sgd_momentum_flat_dynamic against the plain entry: same update with g / scale; a gradient holding inf or NaN leaves w and v
    untouched and halves the scale; `interval` clean steps double it."""
    from dan_amd._lib import call, ptr, stream
    g = torch.Generator().manual_seed(3)
    n = 64 * 40
    w0 = torch.randn(n, generator=g).to(dev)
    gr = torch.randn(n, generator=g).to(dev)
    seg = torch.tensor([0, 64 * 10, n], dtype=torch.int64, device=dev)
    gm = torch.tensor([1.0, 2.0], device=dev)
    wd = torch.tensor([5e-4, 0.0], device=dev)

    def plain(w, v, grad, scale):
        call("danhip_sgd_momentum_flat", ptr(w), ptr(grad), ptr(v), ptr(seg), ptr(gm), ptr(wd), 2, n, 0.1, 0.9, 1.0 / scale, None, stream())

    def dyn(w, v, grad, state):
        call("danhip_sgd_momentum_flat_dynamic", ptr(w), ptr(grad), ptr(v), ptr(seg), ptr(gm), ptr(wd), 2, n, 0.1, 0.9, ptr(state), None, stream())

    state = torch.tensor([256.0, 0.0, 2.0, 0.0], device=dev)
    wa, va, wb, vb = w0.clone(), torch.zeros_like(w0), w0.clone(), torch.zeros_like(w0)
    plain(wa, va, gr * 256.0, 256.0)
    dyn(wb, vb, gr * 256.0, state)
    torch.cuda.synchronize()
    assert torch.equal(wa, wb) and torch.equal(va, vb)
    assert state.tolist() == [256.0, 1.0, 2.0, 0.0]
    dyn(wb, vb, gr * 256.0, state)                       # second clean step: interval reached -> scale doubles
    assert state.tolist() == [512.0, 0.0, 2.0, 0.0]
    for bad in (float("inf"), float("nan"), float("-inf")):
        wk, vk = wb.clone(), vb.clone()
        gbad = (gr * 512.0).clone()
        gbad[n - 3] = bad
        before = state[0].item()
        dyn(wb, vb, gbad, state)
        torch.cuda.synchronize()
        assert torch.equal(wb, wk) and torch.equal(vb, vk), bad      # skipped
        assert state.tolist() == [before * 0.5, 0.0, 2.0, 0.0], (bad, state.tolist())


def test_trainer_with_dynamic_loss_scale(dev):
    """DetectorTrainer(dynamic_loss_scale=True): the device scale multiplies the loss gradients and cancels in the optimizer (same
    trajectory as the static run), grows after the interval, and an overflowing scale costs one skipped step, not the run."""
    from dan_amd import synthetic
    from dan_amd.train_sfd import AnchorConfig, SFDModel, SFDTrainer
    imgs = synthetic.make_images(2, 128, 128, dev, seed=51)
    gts = synthetic.make_gt_boxes(2, 128, 128, seed=52, max_faces=5)
    anchors = AnchorConfig(128, 128, dev)
    loc_t, cls_t, _ = anchors.encode_batch(gts)
    ref = SFDTrainer(SFDModel(device=dev, seed=8), world=1, loss_scale=1.0)
    tr = SFDTrainer(SFDModel(device=dev, seed=8), world=1, loss_scale=256.0, dynamic_loss_scale=True, loss_scale_growth_interval=2)
    for _ in range(2):
        ref.train_step(imgs, loc_t, cls_t)
        tr.train_step(imgs, loc_t, cls_t)
    assert (ref.flat.w - tr.flat.w).abs().max().item() <= 1e-3 * ref.flat.w.abs().max().item()
    assert tr.ls_state.tolist() == [512.0, 0.0, 2.0, 0.0]
    tr.ls_state[0] = float(2.0 ** 126)                   # the scaled head gradients overflow fp32 -> inf in the weight gradients
    w_before = tr.flat.w.clone()
    tr.train_step(imgs, loc_t, cls_t)
    assert torch.equal(tr.flat.w, w_before) and tr.ls_state[0].item() == float(2.0 ** 125)
    tr.ls_state[0] = 512.0
    tr.train_step(imgs, loc_t, cls_t)
    assert not torch.equal(tr.flat.w, w_before) and torch.isfinite(tr.flat.w).all()


def test_batch_without_any_face_trains_on_the_negatives_only(dev):
    """Edge case of the input pipeline (anchor_manipulator.py:286 substitutes a dummy box for an empty list; train_sfd.py:350-417 divides
    by max(n_pos, 1)): a batch whose images hold no face still gives finite losses and gradients, zero localisation loss, and a step."""
    from dan_amd import synthetic
    from dan_amd.train_sfd import AnchorConfig, SFDModel, SFDTrainer
    imgs = synthetic.make_images(2, 128, 128, dev, seed=61)
    anchors = AnchorConfig(128, 128, dev)
    empty = [torch.zeros((0, 4), device=dev), torch.zeros((0, 4), device=dev)]
    loc_t, cls_t, _ = anchors.encode_batch(empty)
    assert (cls_t == 1).sum().item() <= 2 * 6            # at most the dummy [0,0,1,1] box's compensation matches
    tr = SFDTrainer(SFDModel(device=dev, seed=9), world=1)
    w0 = tr.flat.w.clone()
    tr.train_step(imgs, loc_t, cls_t)
    lv = tr.loss_values()
    assert all(v == v and abs(v) < 1e4 for v in (lv["total"], lv["face"][0], lv["face"][1])), lv
    assert torch.isfinite(tr.flat.g).all() and torch.isfinite(tr.flat.w).all() and not torch.equal(tr.flat.w, w0)


def test_graph_captured_dan_step_advances_the_routing_stream(dev):
    from dan_amd import synthetic
    from dan_amd.train_dan import DANModel, DANTrainer, dan_anchor_config, encode_batch_dan
    imgs = synthetic.make_images(2, 128, 128, dev, seed=21)
    gts = synthetic.make_gt_boxes(2, 128, 128, seed=22, max_faces=5)
    anchors = dan_anchor_config(128, 128, dev)
    tr = DANTrainer(DANModel(device=dev), anchors, world=1)
    targets = encode_batch_dan(anchors, gts)
    tr.enable_graph(imgs, *targets, warmup=1)
    per_step = 2 * anchors.num_anchors
    assert int(tr._routing_ctr.item()) == per_step          # the warm-up step ran, recording the graph did not
    masks = []
    for _ in range(3):
        tr.train_step(imgs, *targets)
        masks.append(tr.last_routing[0].clone())
        lv = tr.loss_values()
        assert lv["total"] == lv["total"] and lv["total"] < 1e4
    assert int(tr._routing_ctr.item()) == 4 * per_step      # the device-resident counter moved with every replay


def test_resume_from_own_checkpoint_continues_the_trajectory(dev, tmp_path):
    """DetectorTrainer.save / restore (the Estimator's checkpoint + resume, train_dan.py:549-556): a fresh trainer restored from a
    checkpoint taken after 2 steps predicts identically at once (cached packings of the fused head blocks refreshed) and after one more
    step holds the same weights, momenta and step counter as the run that never stopped (fp32-atomics noise only)."""
    from dan_amd import synthetic
    from dan_amd.train_sfd import AnchorConfig, SFDModel, SFDTrainer
    imgs = synthetic.make_images(2, 128, 128, dev, seed=71)
    gts = synthetic.make_gt_boxes(2, 128, 128, seed=72, max_faces=5)
    anchors = AnchorConfig(128, 128, dev)
    loc_t, cls_t, _ = anchors.encode_batch(gts)
    a = SFDTrainer(SFDModel(device=dev, seed=10), world=1, lr_boundaries=(1, 80000, 100000))       # the schedule moves at step 2
    for _ in range(2):
        a.train_step(imgs, loc_t, cls_t)
    prefix = str(tmp_path / "model.ckpt-2")
    a.save(prefix, "sfd")
    b = SFDTrainer(SFDModel(device=dev, seed=11), world=1, lr_boundaries=(1, 80000, 100000))
    with torch.no_grad():
        before = b.model.forward(imgs)[1].clone()                      # populates b's packed-weight cache with the OLD weights
    got = b.restore(str(tmp_path), "sfd")
    assert "global_step" in got and b.step_no == 2
    with torch.no_grad():
        pa, pb_ = a.model.forward(imgs), b.model.forward(imgs)
    assert torch.equal(pa[0], pb_[0]) and torch.equal(pa[1], pb_[1]) and not torch.equal(before, pb_[1])
    a.train_step(imgs, loc_t, cls_t)
    b.train_step(imgs, loc_t, cls_t)
    scale = a.flat.w.abs().max().item()
    assert (a.flat.w - b.flat.w).abs().max().item() <= 1e-5 * scale
    assert (a.flat.v - b.flat.v).abs().max().item() <= 1e-4 * a.flat.v.abs().max().item()
    assert a.step_no == b.step_no == 3


@pytest.mark.parametrize("which", ["sfd", "dan_deform"])
def test_towers_step_sums_the_towers_gradients_like_the_reference_places_towers_on_one_device(which, dev):
    """DetectorTrainer.train_step_towers (tf_replicate_model_fn.py:504-560 loops the towers, :297-343 sums their gradients, :615-631 puts
    1 / number_of_towers into every tower's loss): the flat gradient after a 2-tower step equals add_n (oracle.train.dp_step) over the
    gradients one tower at a time produces with world = 2 from the same parameters — every backward kernel ACCUMULATES into the sink —
    and the weights equal the Momentum update of that sum.  bench.py's strong-scaling leg runs on this entry."""
    from dan_amd import ops, synthetic
    S, n = 64, 2
    imgs = synthetic.make_images(2 * n, S, S, dev, seed=41)
    gts = synthetic.make_gt_boxes(2 * n, S, S, seed=42, max_faces=4)
    if which == "sfd":
        from dan_amd.train_sfd import AnchorConfig, SFDModel, SFDTrainer
        anchors = AnchorConfig(S, S, dev)
        loc_t, cls_t, _ = anchors.encode_batch(gts)
        make = lambda world: SFDTrainer(SFDModel(device=dev, seed=9), world=world)
        args_of = lambda sl: (imgs[sl].contiguous(), loc_t[sl].contiguous(), cls_t[sl].contiguous())
    else:
        from dan_amd.train_dan import DANModel, DANTrainer, dan_anchor_config, encode_batch_dan
        anchors = dan_anchor_config(S, S, dev)
        loc_t, cls_t, mgt = encode_batch_dan(anchors, gts)
        make = lambda world: DANTrainer(DANModel(device=dev, seed=9, deform=True), anchors, world=world)
        args_of = lambda sl: (imgs[sl].contiguous(), loc_t[sl].contiguous(), cls_t[sl].contiguous(), mgt[sl].contiguous())
    shards = [slice(0, n), slice(n, 2 * n)]
    one = make(2)                                             # one tower at a time, each scaled by 1 / 2
    w0 = one.flat.w.clone()
    gs = []
    for sl in shards:
        one.flat.w.copy_(w0)
        one.flat.v.zero_()
        ops.WEIGHT_EPOCH += 1
        ops.repack_all()
        one.step_no = 0
        if hasattr(one, "_routing_ctr"):
            one._routing_ctr.zero_()
        one.train_step(*args_of(sl))
        torch.cuda.synchronize()
        gs.append(one.flat.g.clone())
    agg, _ = OT.dp_step(lambda i, scale: (0.0, {"flat": gs[i]}), [0, 1])
    want = agg["flat"]
    tw = make(1)
    assert torch.equal(tw.flat.w, w0)
    tw.train_step_towers([args_of(sl) for sl in shards])
    torch.cuda.synchronize()
    assert tw.towers == 1 and tw.step_no == 1
    scale = want.abs().max().item()
    if which == "sfd":
        assert (tw.flat.g - want).abs().max().item() <= 1e-4 * scale
        seg = tw.flat.seg.tolist()
        mult, wd = torch.ones_like(want), torch.zeros_like(want)
        for k in range(len(seg) - 1):
            mult[seg[k]:seg[k + 1]] = tw.flat.gmult[k]
            wd[seg[k]:seg[k + 1]] = tw.flat.wdc[k]
        upd = w0 - 1e-4 * mult * (tw.flat.g + wd * w0)
        assert torch.allclose(tw.flat.w, upd, rtol=1e-5, atol=1e-7)
    else:
        # DAN's train-mode routing draws from a counter-based stream that the second tower continues: whole-gradient agreement in direction
        cos = torch.nn.functional.cosine_similarity(tw.flat.g, want, dim=0).item()
        assert cos >= 0.98 and torch.isfinite(tw.flat.g).all(), cos


@pytest.mark.parametrize("which", ["sfd", "dan"])
def test_bucketwise_optimizer_equals_the_single_pass_optimizer(which, dev, monkeypatch):
    """Round 5: momentum-SGD and the 16-bit re-packing run bucket by bucket on the buckets' stream while backward continues
    (DetectorTrainer._bucket_opt) instead of as one pass behind it.  Same arithmetic per element (train_sfd.py:419-447: L2 term, bias
    gradient x 2, Momentum), so three steps land on the same parameters and momenta as DANHIP_OPT_OVERLAP=0 - up to the order of the
    weight gradients' fp32 partial sums - with the structured zero places of DAN's 'plus' blocks still exactly zero, and the L2 loss the
    buckets add up equals the single pass's."""
    from dan_amd import synthetic
    S = 64
    imgs = synthetic.make_images(2, S, S, dev, seed=51)
    gts = synthetic.make_gt_boxes(2, S, S, seed=52, max_faces=4)

    def run(flag):
        monkeypatch.setenv("DANHIP_OPT_OVERLAP", flag)
        if which == "sfd":
            from dan_amd.train_sfd import AnchorConfig, SFDModel, SFDTrainer
            anchors = AnchorConfig(S, S, dev)
            loc_t, cls_t, _ = anchors.encode_batch(gts)
            tr = SFDTrainer(SFDModel(device=dev, seed=9))
            args = (imgs, loc_t, cls_t)
        else:
            from dan_amd.train_dan import DANModel, DANTrainer, dan_anchor_config, encode_batch_dan
            anchors = dan_anchor_config(S, S, dev)
            tr = DANTrainer(DANModel(device=dev, seed=9), anchors)
            args = (imgs,) + tuple(encode_batch_dan(anchors, gts))
        assert tr.opt_overlap == (flag == "1")
        l2 = []
        for _ in range(3):
            tr.train_step(*args)
            l2.append(tr.loss_values()["l2"])
        torch.cuda.synchronize()
        zeros = tr.flat.w[tr.flat.struct_zero_idx] if tr.flat.struct_zero_idx is not None else None
        return tr.flat.w.clone(), tr.flat.v.clone(), l2, zeros, len(tr.buckets.bounds)

    w1, v1, l1, z1, nb = run("1")
    w0, v0, l0, z0, _ = run("0")
    assert nb >= 2
    scale = w0.abs().max().item()
    assert (w1 - w0).abs().max().item() <= 1e-5 * scale
    assert (v1 - v0).abs().max().item() <= 1e-4 * v0.abs().max().item()
    for a, b in zip(l1, l0):
        assert abs(a - b) <= 1e-5 * abs(b)
    if z1 is not None:
        assert z1.eq(0).all() and z0.eq(0).all()
