"""Anchor / box arithmetic oracle in numpy float32 — restates utility/anchor_manipulator.py and
utility/bbox_util.py of the reference.  Oracle only (see oracle/__init__.py).

All arithmetic is done in float32 in the operation order of the TF graph the reference builds, so the
HIP kernels can be compared bit-for-bit (transcendentals log/exp excepted: see tests for the declared ulp
tolerance).  KATs from SURVEY §8(c) item 5 are checked in tests/test_oracle_anchors.py.
"""
import math

import numpy as np

F32 = np.float32


def center2point(cy, cx, h, w):
    """AnchorEncoder.center2point — anchor_manipulator.py:125-127."""
    one, two = F32(1.0), F32(2.0)
    return cy - (h - one) / two, cx - (w - one) / two, cy + (h - one) / two, cx + (w - one) / two


def point2center(ymin, xmin, ymax, xmax):
    """AnchorEncoder.point2center — anchor_manipulator.py:129-132."""
    one, two = F32(1.0), F32(2.0)
    h, w = (ymax - ymin + one), (xmax - xmin + one)
    return (ymin + ymax) / two, (xmin + xmax) / two, h, w


def get_anchors_width_height(anchor_scale, extra_anchor_scale, anchor_ratio):
    """anchor_manipulator.py:134-161: python float64 math, then tf.constant(float32)."""
    hs, ws = [], []
    for s in extra_anchor_scale:
        hs.append(s)
        ws.append(s)
    for s in anchor_scale:
        for r in anchor_ratio:
            hs.append(s / math.sqrt(r))
            ws.append(s * math.sqrt(r))
    return np.asarray(hs, F32), np.asarray(ws, F32), len(hs)


def generate_anchors_by_offset(anchors_h, anchors_w, depth, layer_shape, stride, offset=0.5):
    """anchor_manipulator.py:163-198: row-major (y, x, depth)."""
    lh, lw = layer_shape
    x_on, y_on = np.meshgrid(np.arange(lw), np.arange(lh))
    off_h, off_w = (offset if isinstance(offset, (list, tuple)) else (offset, offset))
    y_img = (y_on.astype(F32) + F32(off_h)) * F32(stride)
    x_img = (x_on.astype(F32) + F32(off_w)) * F32(stride)
    ymin, xmin, ymax, xmax = center2point(y_img[..., None], x_img[..., None], anchors_h, anchors_w)
    return [a.reshape(-1, depth).astype(F32) for a in (ymin, xmin, ymax, xmax)]


def get_all_anchors(image_shape, anchors_h, anchors_w, depths, offsets, layer_shapes, strides, borders, clips):
    """anchor_manipulator.py:213-273: returns ymin,xmin,ymax,xmax [A] f32 and inside_mask [A] bool."""
    ih, iw = F32(image_shape[0]), F32(image_shape[1])
    cols = [[], [], [], []]
    bord = []
    for i, d in enumerate(depths):
        a = generate_anchors_by_offset(anchors_h[i], anchors_w[i], d, layer_shapes[i], strides[i], offsets[i])
        if clips[i]:
            a = [np.clip(a[0], F32(0), ih - F32(1)), np.clip(a[1], F32(0), iw - F32(1)),
                 np.clip(a[2], F32(0), ih - F32(1)), np.clip(a[3], F32(0), iw - F32(1))]
        for c, v in zip(cols, a):
            c.append(v.reshape(-1))
        bord.append(np.ones_like(a[0].reshape(-1), dtype=F32) * F32(borders[i]))
    ymin, xmin, ymax, xmax = [np.concatenate(c) for c in cols]
    b = np.concatenate(bord)
    inside = (ymin > -b) & (xmin > -b) & (ymax < (ih - F32(1) + b)) & (xmax < (iw - F32(1) + b))
    return ymin, xmin, ymax, xmax, inside


def iou_matrix(boxes_a, boxes_b):
    """areas/intersection/iou_matrix — anchor_manipulator.py:24-52: [A,4] x [G,4] -> [A,G], +1 convention."""
    one = F32(1.0)
    a = boxes_a.astype(F32)
    g = boxes_b.astype(F32)
    ymin, xmin, ymax, xmax = [a[:, i:i + 1] for i in range(4)]
    gymin, gxmin, gymax, gxmax = [g[:, i][None, :] for i in range(4)]
    h = np.maximum(np.minimum(ymax, gymax) - np.maximum(ymin, gymin) + one, F32(0))
    w = np.maximum(np.minimum(xmax, gxmax) - np.maximum(xmin, gxmin) + one, F32(0))
    inter = h * w
    area_a = (xmax - xmin + one) * (ymax - ymin + one)
    area_g = (gxmax - gxmin + one) * (gymax - gymin + one)
    union = area_a + area_g - inter
    with np.errstate(divide="ignore", invalid="ignore"):
        iou = np.where(union == 0, F32(0), inter / union)
    return iou.astype(F32)


def do_dual_max_match(ov, low, high, ignore_between=True, gt_max_first=True):
    """anchor_manipulator.py:54-105.  ov [A,G] f32 -> (match_indices int64 [A], scores f32 [A])."""
    assert gt_max_first
    ov = ov.astype(F32)
    a2g = ov.argmax(1)
    mv = ov.max(1)
    less = mv < F32(low)
    between = (mv < F32(high)) & (mv >= F32(low))
    neg, ign = (less, between) if ignore_between else (between, less)
    idx = np.where(neg, -1, a2g)
    idx = np.where(ign, -2, idx)
    colmax = ov.max(0, keepdims=True)
    left = (ov == colmax)
    left_scores = ov * left.astype(F32)
    any_left = left.max(1) > 0
    pick = np.where(any_left, left_scores.argmax(1), a2g)
    scores = ov[np.arange(ov.shape[0]), pick]
    return np.where(any_left, left_scores.argmax(1), idx).astype(np.int64), scores.astype(F32)


def encode_from_match(bboxes, anchors, matched_gt, prior_scaling, scale=1.0, match_anchor_scale=None):
    """Tail of encode_anchors / encode_pa_anchors (anchor_manipulator.py:294-326, :358-387) given the match.
    anchors = (ymin,xmin,ymax,xmax) ORIGINAL anchors.  Returns targets [A,4] f32, labels [A] i64, matched boxes."""
    ymin, xmin, ymax, xmax = anchors
    mask = matched_gt > -1
    mi = np.clip(matched_gt, 0, np.iinfo(np.int32).max)
    labels = mask.astype(np.int64) + (-1 * (matched_gt < -1).astype(np.int64))
    mb = bboxes.astype(F32)[mi]
    gcy, gcx, gh, gw = point2center(mb[:, 0], mb[:, 1], mb[:, 2], mb[:, 3])
    acy, acx, ah, aw = point2center(ymin, xmin, ymax, xmax)
    ps = [F32(p) for p in prior_scaling]
    t_cy = (gcy - acy) / ah / ps[0]
    t_cx = (gcx - acx) / aw / ps[1]
    if scale == 1.0 and match_anchor_scale is None:
        t_h = np.log(gh / ah) / ps[2]
        t_w = np.log(gw / aw) / ps[3]
    else:
        t_h = np.log(gh * F32(scale) / ah) / ps[2]
        t_w = np.log(gw * F32(scale) / aw) / ps[3]
    tg = np.stack([t_cy, t_cx, t_h, t_w], -1).astype(F32)
    tg = mask.astype(F32)[:, None] * tg
    return tg, labels, (mb * mask.astype(F32)[:, None]).astype(F32)


def encode_anchors(bboxes, anchors, inside_mask, ignore_thr, pos_thr, prior_scaling, match_fn):
    """AnchorEncoder.encode_anchors — anchor_manipulator.py:275-326.
    match_fn(ov) -> (matched_gt, scores): small_mining_match (match_mining=True) or do_dual_max_match."""
    ymin, xmin, ymax, xmax = anchors
    all_a = np.stack([ymin, xmin, ymax, xmax], -1)
    if bboxes.shape[0] < 1:
        bboxes = np.asarray([[0., 0., 1., 1.]], F32)
    ov = iou_matrix(all_a, bboxes) * inside_mask.astype(F32)[:, None]
    matched, scores = match_fn(ov)
    tg, labels, mb = encode_from_match(bboxes, anchors, np.asarray(matched, np.int64), prior_scaling)
    return tg, labels, scores.astype(F32), mb


def encode_pa_anchors(bboxes, anchors, inside_mask, prior_scaling, match_fn, scale):
    """AnchorEncoder.encode_pa_anchors — anchor_manipulator.py:328-387: anchors shrunk by `scale` for matching,
    targets log(gt*scale/anchor)."""
    ymin, xmin, ymax, xmax = anchors
    acy, acx, ah, aw = point2center(ymin, xmin, ymax, xmax)
    s = center2point(acy, acx, ah / F32(scale), aw / F32(scale))
    all_a = np.stack(s, -1).astype(F32)
    if bboxes.shape[0] < 1:
        bboxes = np.asarray([[0., 0., 1., 1.]], F32)
    ov = iou_matrix(all_a, bboxes) * inside_mask.astype(F32)[:, None]
    matched, scores = match_fn(ov)
    tg, labels, mb = encode_from_match(bboxes, anchors, np.asarray(matched, np.int64), prior_scaling, scale=scale,
                                       match_anchor_scale=scale)
    return tg, labels, scores.astype(F32), mb


def decode_anchors(pred, anchors, prior_scaling):
    """decode_anchors / batch_decode_anchors — anchor_manipulator.py:389-424. pred [...,A,4] f32."""
    ymin, xmin, ymax, xmax = anchors
    acy, acx, ah, aw = point2center(ymin, xmin, ymax, xmax)
    ps = [F32(p) for p in prior_scaling]
    pred = pred.astype(F32)
    ph = np.exp(pred[..., 2] * ps[2]) * ah
    pw = np.exp(pred[..., 3] * ps[3]) * aw
    pcy = pred[..., 0] * ps[0] * ah + acy
    pcx = pred[..., 1] * ps[1] * aw + acx
    return np.stack(center2point(pcy, pcx, ph, pw), -1).astype(F32)


# ------------------------------------------------------------------------------- bbox_util
def nms_tf(boxes, scores, max_out, iou_thr):
    """tf.image.non_max_suppression as called from bbox_util.py:77,82 (TF source not vendored; parity
    unpinned): greedy by score desc (stable: ties -> lower index first), suppress when IoU > thr,
    IoU on raw areas (no +1), box corners normalised with min/max."""
    b = boxes.astype(F32)
    order = np.argsort(-scores.astype(F32), kind="stable")
    keep = []
    y1 = np.minimum(b[:, 0], b[:, 2]); y2 = np.maximum(b[:, 0], b[:, 2])
    x1 = np.minimum(b[:, 1], b[:, 3]); x2 = np.maximum(b[:, 1], b[:, 3])
    area = (y2 - y1) * (x2 - x1)
    for i in order:
        if len(keep) >= max_out:
            break
        ok = True
        for j in reversed(keep):                       # TF checks against kept boxes, most recent first
            if area[i] <= 0 or area[j] <= 0:
                iou = F32(0)
            else:
                ih = max(min(y2[i], y2[j]) - max(y1[i], y1[j]), F32(0))
                iw = max(min(x2[i], x2[j]) - max(x1[i], x1[j]), F32(0))
                inter = F32(ih * iw)
                iou = F32(inter / F32(area[i] + area[j] - inter))
            if iou > F32(iou_thr):
                ok = False
                break
        if ok:
            keep.append(int(i))
    return np.asarray(keep, np.int64)


def softmax_np(logits):
    """tf.nn.softmax over the last axis in float32 (max-subtracted, as TF/Eigen does)."""
    x = logits.astype(F32)
    e = np.exp(x - x.max(-1, keepdims=True))
    return (e / e.sum(-1, keepdims=True)).astype(F32)


def parse_by_class(cls_logits, bboxes_pred, image_shape, select_thr, min_size, keep_topk, nms_topk, nms_thr):
    """bbox_util.parse_by_class — bbox_util.py:103-119 for the single 'face' class (num_classes=2):
    softmax -> select(score>thr) -> clip -> filter(min size) -> sort(top_k, zero-padded to keep_topk) ->
    NMS (zero-padded to nms_topk).  Returns (boxes [nms_topk,4], scores [nms_topk])."""
    scores = softmax_np(cls_logits)[:, 1]
    b = bboxes_pred.astype(F32)
    sel = (scores > F32(select_thr)).astype(F32)                     # select_bboxes :24-35
    scores = scores * sel
    b = b * sel[:, None]
    ih, iw = F32(image_shape[0]), F32(image_shape[1])                # clip_bboxes :37-47
    ymin = np.maximum(b[:, 0], F32(0)); xmin = np.maximum(b[:, 1], F32(0))
    ymax = np.minimum(b[:, 2], ih - F32(1)); xmax = np.minimum(b[:, 3], iw - F32(1))
    ymin = np.minimum(ymin, ymax); xmin = np.minimum(xmin, xmax)
    w = xmax - xmin + F32(1)                                         # filter_bboxes :49-59
    h = ymax - ymin + F32(1)
    fm = ((w > F32(min_size) + F32(1)) & (h > F32(min_size) + F32(1))).astype(F32)
    scores = scores * fm
    b = np.stack([ymin * fm, xmin * fm, ymax * fm, xmax * fm], -1)
    k = min(keep_topk, scores.shape[0])                              # sort_bboxes :61-73 (top_k: ties -> lower index)
    order = np.argsort(-scores, kind="stable")[:k]
    scores = np.pad(scores[order], (0, max(keep_topk - k, 0)))
    b = np.pad(b[order], ((0, max(keep_topk - k, 0)), (0, 0)))
    keep = nms_tf(b, scores, nms_topk, nms_thr)                      # nms_bboxes_with_padding :80-91
    ob = np.zeros((nms_topk, 4), F32); os_ = np.zeros((nms_topk,), F32)
    ob[:len(keep)] = b[keep]; os_[:len(keep)] = scores[keep]
    return ob, os_
