#!/usr/bin/env python3
"""Freezes the oracle: writes one seeded input / oracle-output pair per op family to tests/golden/<family>.npz (<= 100 kB each).

    python tests/golden/make_golden.py            # regenerate every fixture (only when the ORACLE is changed on purpose)

SURVEY.md section 8(c), "Fixtures to commit".  The reference holds no output vectors and cannot be run here (no TensorFlow), so these
files do not pin the oracle against the reference — they pin it against ITSELF over time: tests/test_golden_cpu.py recomputes every case
with the current oracle/ and compares bit-for-bit (index / integer outputs) or to 1e-6 (floating point), so the oracle and the HIP kernels
cannot drift together unnoticed.  Inputs are stored beside the outputs: the fixtures do not depend on a generator's stream staying stable.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import anchors as OA      # noqa: E402
from oracle import deform as OD       # noqa: E402
from oracle import evalpipe as OV     # noqa: E402
from oracle import extra_lib as OE    # noqa: E402
from oracle import tf_ops as T        # noqa: E402
from oracle import train as OT        # noqa: E402

F32 = np.float32


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def _boxes(rng, n, size):
    """n ground-truth boxes (ymin, xmin, ymax, xmax) of mixed sizes inside a size x size image."""
    c = rng.uniform(0.1, 0.9, (n, 2)) * size
    hw = np.exp(rng.uniform(np.log(6.0), np.log(size / 3.0), (n, 2)))
    b = np.concatenate([c - hw / 2, c + hw / 2], -1)
    return np.clip(b, 0, size - 1).astype(F32)


def _anchors(size=128):
    """The S3FD anchor pyramid of utility/anchor_manipulator.py on a small image (strides 4..128, scales 16..512)."""
    strides = [4, 8, 16, 32, 64, 128]
    scales = [16., 32., 64., 128., 256., 512.]
    hs, ws, ds = [], [], []
    for s in scales:
        h, w, d = OA.get_anchors_width_height((s,), (), (1.,))
        hs.append(h); ws.append(w); ds.append(d)
    shapes = [(-(-size // s), -(-size // s)) for s in strides]
    out = OA.get_all_anchors((size, size), hs, ws, ds, [0.5] * 6, shapes, strides, [0.] * 6, [False] * 6)
    return out


# --------------------------------------------------------------------------------------------------------------- cases
# Each case: name -> (inputs dict, compute(inputs) -> outputs dict, {"exact": [...]}): `exact` outputs are compared bit-for-bit.
def case_conv():
    rng = np.random.default_rng(101)
    x = rng.standard_normal((2, 9, 11, 8)).astype(F32)
    w3 = (rng.standard_normal((3, 3, 8, 16)) * 0.2).astype(F32)
    w1 = (rng.standard_normal((1, 1, 8, 16)) * 0.3).astype(F32)
    b = rng.standard_normal(16).astype(F32)
    dy = rng.standard_normal((2, 9, 11, 16)).astype(F32)
    inp = dict(x=x, w3=w3, w1=w1, b=b, dy=dy)

    def compute(i):
        xs = _t(i["x"]).requires_grad_(True)
        ws = _t(i["w3"]).requires_grad_(True)
        bs = _t(i["b"]).requires_grad_(True)
        y = T.conv2d_same(xs, ws, bs, 1, relu=True)
        gx, gw, gb = torch.autograd.grad(y, [xs, ws, bs], _t(i["dy"]))
        y2 = T.conv2d_same(_t(i["x"]), _t(i["w3"]), _t(i["b"]), 2, relu=False)       # stride 2: asymmetric SAME padding
        y1 = T.conv2d_same(_t(i["x"]), _t(i["w1"]), None, 1, relu=False)
        yv = T.conv2d_valid(_t(i["x"]), _t(i["w3"]), _t(i["b"]), 2, relu=True)
        return dict(y_s1_relu=y.detach().numpy(), dx=gx.numpy(), dw=gw.numpy(), db=gb.numpy(), y_s2=y2.numpy(), y_1x1=y1.numpy(), y_valid_s2=yv.numpy())
    return inp, compute, {"exact": []}


def case_pools_resize_l2norm():
    rng = np.random.default_rng(102)
    x = rng.standard_normal((2, 7, 9, 8)).astype(F32)
    gamma = rng.uniform(5, 12, 8).astype(F32)
    lat = rng.standard_normal((2, 13, 18, 8)).astype(F32)
    cls = rng.standard_normal((1, 4, 5, 4)).astype(F32)
    inp = dict(x=x, gamma=gamma, lat=lat, cls=cls)

    def compute(i):
        xs = _t(i["x"]).requires_grad_(True)
        mp = T.max_pool_2x2_same(xs)
        gmp, = torch.autograd.grad(mp, xs, torch.ones_like(mp))
        ln = T.l2_normalize(xs, _t(i["gamma"]))
        gln, = torch.autograd.grad(ln.sum() * 0.5 + (ln * ln).sum(), xs)
        rs = T.resize_bilinear_legacy(xs, 13, 18) + _t(i["lat"])
        grs, = torch.autograd.grad((rs * _t(i["lat"])).sum(), xs)
        xd = xs.detach()
        return dict(maxpool=mp.detach().numpy(), maxpool_grad_ones=gmp.numpy(), avgpool=T.avg_pool_2x2_s1_same(xd).numpy(),
                    maxpool3x3s2=T.max_pool_3x3_s2_same(xd).numpy(), l2norm=ln.detach().numpy(), l2norm_grad=gln.numpy(),
                    resize_add=rs.detach().numpy(), resize_grad=grs.numpy(), maxout=T.maxout_cls(_t(i["cls"]), 1, 3, 1).numpy())
    return inp, compute, {"exact": ["maxpool", "maxpool_grad_ones", "maxpool3x3s2", "maxout"]}


def case_deform():
    rng = np.random.default_rng(103)
    B, C, H, W, Co, dg = 1, 8, 7, 8, 6, 2
    x = rng.standard_normal((B, C, H, W)).astype(F32)
    w = (rng.standard_normal((Co, C, 3, 3)) * 0.2).astype(F32)
    off = (rng.standard_normal((B, 2 * 9 * dg, H, W)) * 2.0).astype(F32)          # sigma = 2 px: most samples leave their cell, some the image
    dy = rng.standard_normal((B, Co, H, W)).astype(F32)
    inp = dict(x=x, w=w, off=off, dy=dy, dg=np.int32(dg))

    def compute(i):
        g = int(i["dg"])
        y = OD.deform_conv_forward(_t(i["x"]), _t(i["w"]), _t(i["off"]), 1, 1, g)
        dx, dw, doff = OD.deform_conv_backward(_t(i["x"]), _t(i["w"]), _t(i["off"]), _t(i["dy"]), 1, 1, g)
        col = OD.deform_im2col(_t(i["x"]), _t(i["off"]), 3, 3, 1, 1, g)
        return dict(y=y.numpy(), dx=dx.numpy(), dw=dw.numpy(), doff=doff.numpy(), col=col.numpy())
    return inp, compute, {"exact": []}


def case_matching():
    rng = np.random.default_rng(104)
    anchors = _anchors(128)
    ymin, xmin, ymax, xmax, inside = anchors
    gts = _boxes(rng, 7, 128)
    all_a = np.stack([ymin, xmin, ymax, xmax], -1)
    ov = (OA.iou_matrix(all_a, gts) * inside.astype(F32)[:, None]).astype(F32)
    inp = dict(gts=gts, ov=ov)

    def compute(i):
        o = i["ov"]
        dm_i, dm_s = OA.do_dual_max_match(o, 0.35, 0.35)
        dm2_i, dm2_s = OA.do_dual_max_match(o, 0.1, 0.5, ignore_between=False)
        sm_i, sm_s = OE.small_mining_match(o, 0.0, 0.35, 0.35, 6, 0.1)
        return dict(iou=OA.iou_matrix(all_a, i["gts"]).astype(F32), dual_idx=dm_i, dual_scores=dm_s, dual2_idx=dm2_i, dual2_scores=dm2_s,
                    mining_idx=sm_i, mining_scores=sm_s)
    return inp, compute, {"exact": ["dual_idx", "dual2_idx", "mining_idx", "dual_scores", "dual2_scores", "mining_scores"]}


def case_encode_decode():
    rng = np.random.default_rng(105)
    ymin, xmin, ymax, xmax, inside = _anchors(128)
    gts = _boxes(rng, 5, 128)
    pred = (rng.standard_normal((ymin.shape[0], 4)) * 0.5).astype(F32)
    inp = dict(gts=gts, pred=pred)
    A = (ymin, xmin, ymax, xmax)
    ps = (0.1, 0.1, 0.2, 0.2)

    def compute(i):
        mine = lambda o: OE.small_mining_match(o, 0.0, 0.35, 0.35, 6, 0.1)
        dual = lambda o: OA.do_dual_max_match(o, 0.35, 0.35)
        t1, l1, s1, _ = OA.encode_anchors(i["gts"], A, inside, 0.35, 0.35, ps, mine)
        t2, l2, s2, _ = OA.encode_anchors(i["gts"], A, inside, 0.35, 0.35, ps, dual)
        t3, l3, s3, _ = OA.encode_pa_anchors(i["gts"], A, inside, ps, mine, 2.0)
        t4, l4, s4, _ = OA.encode_pa_anchors(i["gts"], A, inside, ps, dual, 4.0)
        t5, l5, s5, _ = OA.encode_anchors(np.zeros((0, 4), F32), A, inside, 0.35, 0.35, ps, mine)        # empty image
        return dict(anchors=np.stack(A, -1), inside=inside.astype(np.int32), enc_mining_t=t1, enc_mining_l=l1, enc_mining_s=s1,
                    enc_dual_t=t2, enc_dual_l=l2, pa2_t=t3, pa2_l=l3, pa4_t=t4, pa4_l=l4, empty_t=t5, empty_l=l5,
                    decoded=OA.decode_anchors(i["pred"], A, ps))
    return inp, compute, {"exact": ["anchors", "inside", "enc_mining_l", "enc_dual_l", "pa2_l", "pa4_l", "empty_l", "empty_t",
                                    "enc_mining_t", "enc_dual_t", "pa2_t", "pa4_t", "enc_mining_s", "decoded"]}


def case_routing():
    rng = np.random.default_rng(106)
    fh, fw, depth, stride = 8, 8, 1, 8
    N = fh * fw * depth
    cy, cx = np.meshgrid(np.arange(fh), np.arange(fw), indexing="ij")
    ctr = np.stack([cy, cx], -1).reshape(-1, 2) * stride + stride / 2.0
    jit = rng.uniform(-12, 12, (N, 2))
    hw = rng.uniform(6, 30, (N, 2))
    anchors = np.concatenate([ctr + jit - hw / 2, ctr + jit + hw / 2], -1).astype(F32)       # refined (decoded) boxes of stage 1
    gt = np.tile(_boxes(rng, 1, 64), (N, 1)).astype(F32)
    labels = rng.uniform(0, 1, N).astype(F32)
    mask_in = (rng.uniform(0, 1, N) > 0.2).astype(np.int32)
    u = rng.uniform(0, 1, N)
    inp = dict(anchors=anchors, gt=gt, labels=labels, mask_in=mask_in, u=u)

    def compute(i):
        me, de = OE.dynamic_anchor_routing(i["anchors"], i["gt"], i["labels"], i["mask_in"], fh, fw, depth, stride, 64, 64, False, 0.03, 0.0)
        mt, dt = OE.dynamic_anchor_routing(i["anchors"], i["gt"], i["labels"], i["mask_in"], fh, fw, depth, stride, 64, 64, True, 0.4, 0.35, u=i["u"])
        return dict(eval_mask=me, eval_decode=de, train_mask=mt, train_decode=dt, uniform_stream=OE.uniform_stream(12345, 7, 16))
    return inp, compute, {"exact": ["eval_mask", "eval_decode", "train_mask", "train_decode", "uniform_stream"]}


def case_nms_vote():
    rng = np.random.default_rng(107)
    base = _boxes(rng, 12, 200)
    boxes = np.concatenate([base + rng.uniform(-3, 3, base.shape).astype(F32) for _ in range(6)], 0).astype(F32)
    scores = rng.uniform(0.02, 1.0, boxes.shape[0]).astype(F32)
    scores[5] = scores[9]                                                            # a tie: lower index first
    logits = (rng.standard_normal((boxes.shape[0], 2)) * 2).astype(F32)
    det = np.concatenate([boxes[:, [1, 0, 3, 2]], scores[:, None]], -1).astype(F32)   # eval_dan.py rows: xmin, ymin, xmax, ymax, score
    inp = dict(boxes=boxes, scores=scores, logits=logits, det=det)

    def compute(i):
        keep = OA.nms_tf(i["boxes"], i["scores"], 30, 0.3)
        pb, ps = OA.parse_by_class(i["logits"], i["boxes"], (200, 200), 0.1, 4.0, 50, 20, 0.3)
        return dict(nms_keep=keep, parsed_boxes=pb, parsed_scores=ps, softmax=OA.softmax_np(i["logits"]), voted=OV.bbox_vote(i["det"].astype(np.float64)))
    return inp, compute, {"exact": ["nms_keep", "parsed_boxes", "voted"]}


def case_loss():
    rng = np.random.default_rng(108)
    B, A = 3, 400
    cls = (rng.standard_normal((B, A, 2)) * 1.5).astype(F32)
    loc = rng.standard_normal((B, A, 4)).astype(F32)
    loc_t = rng.standard_normal((B, A, 4)).astype(F32)
    lab = np.where(rng.uniform(0, 1, (B, A)) < 0.04, 1, 0).astype(np.int64)
    lab[rng.uniform(0, 1, (B, A)) < 0.05] = -1
    lab[2] = np.where(lab[2] > 0, 0, lab[2])                                         # an image without positives
    inp = dict(cls=cls, loc=loc, loc_t=loc_t, labels=lab)

    def compute(i):
        c = _t(i["cls"]).requires_grad_(True)
        l_ = _t(i["loc"]).requires_grad_(True)
        out = {}
        for tag, alo in (("sfd", False), ("dan", True)):
            ce, ll, final = OT.detection_loss(c, l_, _t(i["labels"]), _t(i["loc_t"]), 3.0, alo)
            gc, gl = torch.autograd.grad(ce + ll, [c, l_])
            out.update({tag + "_ce": ce.detach().numpy(), tag + "_loc": ll.detach().numpy(), tag + "_selected": final.numpy(),
                        tag + "_dcls": gc.numpy(), tag + "_dloc": gl.numpy()})
        w = {"a/kernel": _t(i["cls"][0]), "a/bias": _t(i["loc"][0, :, 0])}
        g = {k: v * 0.5 for k, v in w.items()}
        m = {k: torch.zeros_like(v) for k, v in w.items()}
        p = {k: v.clone() for k, v in w.items()}
        OT.momentum_sgd_step(p, g, m, 1e-3)
        OT.momentum_sgd_step(p, g, m, 1e-3)
        out.update(sgd_kernel=p["a/kernel"].numpy(), sgd_bias=p["a/bias"].numpy(), l2=OT.l2_regularizer(w).numpy())
        return out
    return inp, compute, {"exact": ["sfd_selected", "dan_selected"]}


CASES = {"conv": case_conv, "pools_resize_l2norm": case_pools_resize_l2norm, "deform": case_deform, "matching": case_matching,
         "encode_decode": case_encode_decode, "routing": case_routing, "nms_vote": case_nms_vote, "loss": case_loss}


def main():
    torch.set_num_threads(1)
    for name, fn in CASES.items():
        inp, compute, meta = fn()
        out = compute(inp)
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, exact=np.asarray(meta["exact"], dtype="U32"), **{"in_" + k: v for k, v in inp.items()},
                            **{"out_" + k: np.asarray(v) for k, v in out.items()})
        kb = os.path.getsize(path) / 1024.0
        assert kb <= 100.0, (name, kb)
        print("%-22s %6.1f kB  %d outputs" % (name, kb, len(out)))


if __name__ == "__main__":
    main()
