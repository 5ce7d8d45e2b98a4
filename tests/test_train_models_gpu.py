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
    for on in ("1", "0"):
        monkeypatch.setenv("DANHIP_WGRAD_STREAM", on)
        tr = SFDTrainer(SFDModel(device=dev, seed=6), world=1)
        tr.train_step(imgs, loc_t, cls_t)
        torch.cuda.synchronize()
        assert not ops._WGRAD["on"] and not ops._WGRAD["keep"]          # joined, nothing kept alive
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
