"""GPU anchor utilities — mirrors the reference's utility/anchor_manipulator.py operator surface (AnchorEncoder with
get_anchors_width_height / get_anchors_count / get_all_anchors / encode_anchors / encode_pa_anchors /
batch_decode_anchors / decode_anchors; module functions iou_matrix, do_dual_max_match) on device tensors, through the
libdanhip C ABI.  fp32 + int32, bit-exact against oracle/anchors.py (log/exp: see tests).
"""
import math

import torch

from .. import _lib
from .._lib import call, ptr, stream


def _ws(A, G, dev):
    n = _lib.lib().danhip_match_workspace_bytes(A, G)
    return torch.empty((n + 7) // 8, dtype=torch.int64, device=dev), n


def iou_matrix(anchors4, gt_bboxes, inside_mask=None):
    """anchor_manipulator.py:44-52.  anchors4 = (ymin,xmin,ymax,xmax) [A] each; gt_bboxes [G,4] -> [A,G] fp32
    (already multiplied by inside_mask when given, as encode_anchors does at :287)."""
    ymin, xmin, ymax, xmax = [t.contiguous() for t in anchors4]
    A, G = ymin.numel(), gt_bboxes.shape[0]
    ov = torch.empty((A, G), dtype=torch.float32, device=ymin.device)
    im = inside_mask.to(torch.uint8).contiguous() if inside_mask is not None else None
    call("danhip_iou_matrix", ptr(ymin), ptr(xmin), ptr(ymax), ptr(xmax), ptr(im), ptr(gt_bboxes.contiguous()), ptr(ov), A, G, stream())
    return ov


def do_dual_max_match(overlap_matrix, low_thres, high_thres, ignore_between=True, gt_max_first=True):
    """anchor_manipulator.py:54-105 -> (match_indices int32 [A], match_scores fp32 [A])."""
    if not gt_max_first:
        raise NotImplementedError("gt_max_first=False is never used by the reference scripts")
    A, G = overlap_matrix.shape
    idx = torch.empty((A,), dtype=torch.int32, device=overlap_matrix.device)
    sc = torch.empty((A,), dtype=torch.float32, device=overlap_matrix.device)
    ws, n = _ws(A, G, overlap_matrix.device)
    call("danhip_dual_max_match", ptr(overlap_matrix.contiguous()), A, G, float(low_thres), float(high_thres), int(ignore_between), ptr(idx),
         ptr(sc), ptr(ws), n, stream())
    return idx, sc


def small_mining_match(overlaps, negative_low_thres, negative_high_thres, positive_thres, min_match, stop_positive_thres):
    """custom_op.small_mining_match (utility/custom_op.py:48; op def cpp/ExtraLib/small_mining_match.cc:31-54)."""
    A, G = overlaps.shape
    idx = torch.empty((A,), dtype=torch.int32, device=overlaps.device)
    sc = torch.empty((A,), dtype=torch.float32, device=overlaps.device)
    ws, n = _ws(A, G, overlaps.device)
    call("danhip_small_mining_match", ptr(overlaps.contiguous()), A, G, float(negative_low_thres), float(negative_high_thres),
         float(positive_thres), int(min_match), float(stop_positive_thres), ptr(idx), ptr(sc), ptr(ws), n, stream())
    return idx, sc


class AnchorEncoder(object):
    def __init__(self, positive_threshold, ignore_threshold, prior_scaling, device="cuda"):
        self._positive_threshold = positive_threshold
        self._ignore_threshold = ignore_threshold
        self._prior_scaling = [float(p) for p in prior_scaling]
        self.device = torch.device(device)

    def get_anchors_width_height(self, anchor_scale, extra_anchor_scale, anchor_ratio, name=None):
        """anchor_manipulator.py:134-161 (python float64 math, rounded once to fp32)."""
        hs, ws = [], []
        for s in extra_anchor_scale:
            hs.append(s)
            ws.append(s)
        for s in anchor_scale:
            for r in anchor_ratio:
                hs.append(s / math.sqrt(r))
                ws.append(s * math.sqrt(r))
        return (torch.tensor(hs, dtype=torch.float32, device=self.device), torch.tensor(ws, dtype=torch.float32, device=self.device), len(hs))

    def get_anchors_count(self, anchors_depth, layer_shape, name=None):
        n = layer_shape[0] * layer_shape[1]
        return n, n * anchors_depth

    def get_all_anchors(self, image_shape, anchors_height, anchors_width, anchors_depth, anchors_offsets, layer_shapes, feat_strides,
                        allowed_borders, should_clips, name=None):
        """anchor_manipulator.py:213-273 -> (ymin, xmin, ymax, xmax, inside_mask[bool])."""
        counts = [int(ls[0]) * int(ls[1]) * d for ls, d in zip(layer_shapes, anchors_depth)]
        A = sum(counts)
        out = [torch.empty((A,), dtype=torch.float32, device=self.device) for _ in range(4)]
        off = 0
        for i, d in enumerate(anchors_depth):
            o = anchors_offsets[i]
            oh, ow = (o if isinstance(o, (list, tuple)) else (o, o))
            call("danhip_anchors_generate", ptr(out[0]), ptr(out[1]), ptr(out[2]), ptr(out[3]), ptr(anchors_height[i]), ptr(anchors_width[i]), d,
                 int(layer_shapes[i][0]), int(layer_shapes[i][1]), float(feat_strides[i]), float(oh), float(ow), off, stream())
            off += counts[i]
        ymin, xmin, ymax, xmax = out
        ih, iw = float(image_shape[0]), float(image_shape[1])
        border = torch.cat([torch.full((c,), float(b), dtype=torch.float32, device=self.device) for c, b in zip(counts, allowed_borders)])
        if any(should_clips):
            clip = torch.cat([torch.full((c,), bool(s), dtype=torch.bool, device=self.device) for c, s in zip(counts, should_clips)])
            ymin = torch.where(clip, ymin.clamp(0.0, ih - 1.0), ymin)
            xmin = torch.where(clip, xmin.clamp(0.0, iw - 1.0), xmin)
            ymax = torch.where(clip, ymax.clamp(0.0, ih - 1.0), ymax)
            xmax = torch.where(clip, xmax.clamp(0.0, iw - 1.0), xmax)
        inside = (ymin > -border) & (xmin > -border) & (ymax < (ih - 1.0 + border)) & (xmax < (iw - 1.0 + border))
        return ymin, xmin, ymax, xmax, inside

    def _encode(self, bboxes, anchors4, match_anchors4, inside_mask, ignore_thr, pos_thr, match_mining, scale):
        if bboxes.shape[0] < 1:
            bboxes = torch.tensor([[0., 0., 1., 1.]], dtype=torch.float32, device=self.device)
        bboxes = bboxes.to(torch.float32).contiguous()
        ov = iou_matrix(match_anchors4, bboxes, inside_mask)
        if match_mining:
            matched, scores = small_mining_match(ov, 0., ignore_thr, pos_thr, 6, 0.3)
        else:
            matched, scores = do_dual_max_match(ov, ignore_thr, pos_thr)
        ymin, xmin, ymax, xmax = anchors4
        A = ymin.numel()
        targets = torch.empty((A, 4), dtype=torch.float32, device=self.device)
        labels = torch.empty((A,), dtype=torch.int32, device=self.device)
        mgt = torch.empty((A, 4), dtype=torch.float32, device=self.device)
        ps = self._prior_scaling
        call("danhip_encode_anchors", ptr(ymin), ptr(xmin), ptr(ymax), ptr(xmax), ptr(bboxes), ptr(matched), ptr(targets), ptr(labels), ptr(mgt), A,
             ps[0], ps[1], ps[2], ps[3], float(scale), stream())
        return targets, labels, scores, mgt

    def _encode_batch(self, bboxes_list, anchors4, match_anchors4, inside_mask, ignore_thr, pos_thr, match_mining, scale):
        """The per-image map of anchor_encoder_fn over a batch as ONE library call (danhip_encode_anchors_batched): the images'
        sequential hard-face compensation passes run side by side.  Row b equals _encode(bboxes_list[b], ...) bit for bit."""
        B = len(bboxes_list)
        rows = []
        for b in bboxes_list:
            b = torch.as_tensor(b, dtype=torch.float32).reshape(-1, 4)
            rows.append(b if b.shape[0] > 0 else torch.tensor([[0., 0., 1., 1.]], dtype=torch.float32, device=b.device))  # :286 / :347
        counts = [int(r.shape[0]) for r in rows]
        offs = [0]
        for c in counts:
            offs.append(offs[-1] + c)
        gt = torch.cat([r.to(self.device) for r in rows]).contiguous()
        goff = torch.tensor(offs, dtype=torch.int32).to(self.device, non_blocking=True)
        ymin, xmin, ymax, xmax = [t.contiguous() for t in anchors4]
        m4 = [None] * 4 if match_anchors4 is anchors4 else [t.contiguous() for t in match_anchors4]
        A = ymin.numel()
        im = inside_mask.to(torch.uint8).contiguous() if inside_mask is not None else None
        targets = torch.empty((B, A, 4), dtype=torch.float32, device=self.device)
        labels = torch.empty((B, A), dtype=torch.int32, device=self.device)
        scores = torch.empty((B, A), dtype=torch.float32, device=self.device)
        mgt = torch.empty((B, A, 4), dtype=torch.float32, device=self.device)
        n = _lib.lib().danhip_encode_anchors_batched_workspace_bytes(B, A, offs[-1])
        ws = torch.empty((n + 7) // 8, dtype=torch.int64, device=self.device)
        ps = self._prior_scaling
        call("danhip_encode_anchors_batched", ptr(ymin), ptr(xmin), ptr(ymax), ptr(xmax), ptr(m4[0]), ptr(m4[1]), ptr(m4[2]), ptr(m4[3]), ptr(im),
             ptr(gt), ptr(goff), B, A, offs[-1], max(counts), int(bool(match_mining)), 0., float(ignore_thr), float(pos_thr), 6, 0.3, ps[0], ps[1],
             ps[2], ps[3], float(scale), ptr(targets), ptr(labels), ptr(scores), ptr(mgt), ptr(ws), n, stream())
        return targets, labels, scores, mgt

    def encode_anchors_batch(self, bboxes_list, anchors_ymin, anchors_xmin, anchors_ymax, anchors_xmax, inside_mask, match_mining=False):
        """encode_anchors (anchor_manipulator.py:275-326) for a list of images -> ([B,A,4], [B,A] int32, [B,A], [B,A,4])."""
        a4 = (anchors_ymin, anchors_xmin, anchors_ymax, anchors_xmax)
        return self._encode_batch(bboxes_list, a4, a4, inside_mask, self._ignore_threshold, self._positive_threshold, match_mining, 1.0)

    def encode_pa_anchors_batch(self, bboxes_list, anchors_ymin, anchors_xmin, anchors_ymax, anchors_xmax, inside_mask, ignore_threshold,
                                positive_threshold, match_mining=True, scale=1.):
        """encode_pa_anchors (anchor_manipulator.py:328-387) for a list of images."""
        a4 = (anchors_ymin, anchors_xmin, anchors_ymax, anchors_xmax)
        return self._encode_batch(bboxes_list, a4, self._pa_match_anchors(a4, scale), inside_mask, ignore_threshold, positive_threshold,
                                  match_mining, scale)

    @staticmethod
    def _pa_match_anchors(a4, scale):
        anchors_ymin, anchors_xmin, anchors_ymax, anchors_xmax = a4
        h = anchors_ymax - anchors_ymin + 1.0
        w = anchors_xmax - anchors_xmin + 1.0
        cy = (anchors_ymin + anchors_ymax) / 2.0
        cx = (anchors_xmin + anchors_xmax) / 2.0
        hs, ws = h / float(scale), w / float(scale)
        return (cy - (hs - 1.0) / 2.0, cx - (ws - 1.0) / 2.0, cy + (hs - 1.0) / 2.0, cx + (ws - 1.0) / 2.0)

    def encode_anchors(self, bboxes, anchors_ymin, anchors_xmin, anchors_ymax, anchors_xmax, inside_mask, match_mining=False, debug=False):
        """anchor_manipulator.py:275-326 -> (gt_targets [A,4], gt_labels [A] int32 in {1,0,-1}, gt_scores [A], matched_gt [A,4])."""
        a4 = (anchors_ymin, anchors_xmin, anchors_ymax, anchors_xmax)
        return self._encode(bboxes, a4, a4, inside_mask, self._ignore_threshold, self._positive_threshold, match_mining, 1.0)

    def encode_pa_anchors(self, bboxes, anchors_ymin, anchors_xmin, anchors_ymax, anchors_xmax, inside_mask, ignore_threshold,
                          positive_threshold, match_mining=True, scale=1., debug=False):
        """anchor_manipulator.py:328-387: anchors shrunk by `scale` around their centre for matching."""
        a4 = (anchors_ymin, anchors_xmin, anchors_ymax, anchors_xmax)
        h = anchors_ymax - anchors_ymin + 1.0
        w = anchors_xmax - anchors_xmin + 1.0
        cy = (anchors_ymin + anchors_ymax) / 2.0
        cx = (anchors_xmin + anchors_xmax) / 2.0
        hs, ws = h / float(scale), w / float(scale)
        m4 = (cy - (hs - 1.0) / 2.0, cx - (ws - 1.0) / 2.0, cy + (hs - 1.0) / 2.0, cx + (ws - 1.0) / 2.0)
        return self._encode(bboxes, a4, m4, inside_mask, ignore_threshold, positive_threshold, match_mining, scale)

    def batch_decode_anchors(self, pred_location, anchors_ymin, anchors_xmin, anchors_ymax, anchors_xmax):
        """anchor_manipulator.py:389-408: pred [B,A,4] -> boxes [B,A,4]."""
        B, A, _ = pred_location.shape
        out = torch.empty((B, A, 4), dtype=torch.float32, device=pred_location.device)
        ps = self._prior_scaling
        call("danhip_decode_anchors", ptr(pred_location.contiguous()), ptr(anchors_ymin), ptr(anchors_xmin), ptr(anchors_ymax), ptr(anchors_xmax),
             ptr(out), B, A, ps[0], ps[1], ps[2], ps[3], stream())
        return out

    def decode_anchors(self, pred_location, anchors_ymin, anchors_xmin, anchors_ymax, anchors_xmax):
        """anchor_manipulator.py:409-424: pred [A,4] -> boxes [A,4]."""
        return self.batch_decode_anchors(pred_location.unsqueeze(0), anchors_ymin, anchors_xmin, anchors_ymax, anchors_xmax)[0]
