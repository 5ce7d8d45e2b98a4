"""Host-side mirror of the reference's utility/bbox_util.py on device tensors: select / clip / filter / sort are exact
fp32 elementwise steps (torch plumbing), the candidate ordering and NMS are libdanhip kernels (tf.nn.top_k / tf.image.non_max_suppression).
Function names, argument order and return structure follow bbox_util.py:24-119."""
import torch

from .._lib import call, ptr, stream


def _order(scores):
    """tf.nn.top_k / non_max_suppression candidate order: descending score, ties -> lower index first (libdanhip arg-sort)."""
    from .. import ops
    return ops.argsort_desc(scores.to(torch.float32))


def select_bboxes(scores_pred, bboxes_pred, num_classes, select_threshold, name=None):
    """bbox_util.py:24-35."""
    selected_bboxes, selected_scores = {}, {}
    for class_ind in range(1, num_classes):
        class_scores = scores_pred[:, class_ind]
        select_mask = (class_scores > select_threshold).to(torch.float32)
        selected_bboxes[class_ind] = bboxes_pred * select_mask.unsqueeze(-1)
        selected_scores[class_ind] = class_scores * select_mask
    return selected_bboxes, selected_scores


def clip_bboxes(ymin, xmin, ymax, xmax, height, width, name=None):
    """bbox_util.py:37-47."""
    ymin = torch.clamp(ymin, min=0.)
    xmin = torch.clamp(xmin, min=0.)
    ymax = torch.clamp(ymax, max=float(height) - 1.)
    xmax = torch.clamp(xmax, max=float(width) - 1.)
    return torch.minimum(ymin, ymax), torch.minimum(xmin, xmax), ymax, xmax


def filter_bboxes(scores_pred, ymin, xmin, ymax, xmax, min_size, name=None):
    """bbox_util.py:49-59."""
    width = xmax - xmin + 1.
    height = ymax - ymin + 1.
    m = ((width > min_size + 1.) & (height > min_size + 1.)).to(torch.float32)
    return scores_pred * m, ymin * m, xmin * m, ymax * m, xmax * m


def sort_bboxes(scores_pred, ymin, xmin, ymax, xmax, keep_topk, name=None):
    """bbox_util.py:61-73: tf.nn.top_k (ties -> lower index first) then zero padding up to keep_topk."""
    n = scores_pred.shape[0]
    k = min(int(keep_topk), n)
    order = _order(scores_pred)[:k]
    pad = max(int(keep_topk) - n, 0)
    f = lambda t: torch.nn.functional.pad(t[order], (0, pad))
    return f(scores_pred), f(ymin), f(xmin), f(ymax), f(xmax)


def _nms_indices(scores_pred, bboxes_pred, nms_topk, nms_threshold):
    order = _order(scores_pred)                                                    # TF orders candidates by score
    b = bboxes_pred[order].to(torch.float32).contiguous()
    K = b.shape[0]
    keep = torch.empty((1, int(nms_topk)), dtype=torch.int32, device=b.device)
    num = torch.empty((1,), dtype=torch.int32, device=b.device)
    call("danhip_nms", ptr(b), 1, K, int(nms_topk), float(nms_threshold), ptr(keep), ptr(num), stream())
    n = int(num.item())
    return order[keep[0, :n].long()]


def nms_bboxes(scores_pred, bboxes_pred, nms_topk, nms_threshold, name=None):
    """bbox_util.py:75-78."""
    idx = _nms_indices(scores_pred, bboxes_pred, nms_topk, nms_threshold)
    return scores_pred[idx], bboxes_pred[idx]


def nms_bboxes_with_padding(scores_pred, bboxes_pred, nms_topk, nms_threshold, name=None):
    """bbox_util.py:80-91 (zero padded to nms_topk)."""
    idx = _nms_indices(scores_pred, bboxes_pred, nms_topk, nms_threshold)
    pad = int(nms_topk) - idx.shape[0]
    return torch.nn.functional.pad(scores_pred[idx], (0, pad)), torch.nn.functional.pad(bboxes_pred[idx], (0, 0, 0, pad))


def bbox_point2center(bboxes, name=None):
    """bbox_util.py:93-97."""
    ymin, xmin, ymax, xmax = bboxes.unbind(-1)
    return torch.stack([(ymin + ymax) / 2., (xmin + xmax) / 2., ymax - ymin + 1., xmax - xmin + 1.], dim=-1)


def bbox_center2point(bboxes, name=None):
    """bbox_util.py:99-102."""
    y, x, h, w = bboxes.unbind(-1)
    return torch.stack([y - (h - 1.) / 2., x - (w - 1.) / 2., y + (h - 1.) / 2., x + (w - 1.) / 2.], dim=-1)


def parse_by_class(image_shape, cls_pred, bboxes_pred, num_classes, select_threshold, min_size, keep_topk, nms_topk, nms_threshold):
    """bbox_util.py:103-119 -> ({class: boxes [nms_topk,4]}, {class: scores [nms_topk]})."""
    if cls_pred.shape[-1] == 2:                      # every graph here: background / face -> one libdanhip launch for the face column
        from .. import ops
        p = ops.face_scores(cls_pred.to(torch.float32))
        scores_pred = torch.stack([1. - p, p], dim=-1)        # (class 0 is never selected: the loop below starts at 1)
    else:
        scores_pred = torch.softmax(cls_pred.to(torch.float32), dim=-1)
    selected_bboxes, selected_scores = select_bboxes(scores_pred, bboxes_pred.to(torch.float32), num_classes, select_threshold)
    for c in range(1, num_classes):
        ymin, xmin, ymax, xmax = selected_bboxes[c].unbind(-1)
        ymin, xmin, ymax, xmax = clip_bboxes(ymin, xmin, ymax, xmax, image_shape[0], image_shape[1])
        sc, ymin, xmin, ymax, xmax = filter_bboxes(selected_scores[c], ymin, xmin, ymax, xmax, min_size)
        sc, ymin, xmin, ymax, xmax = sort_bboxes(sc, ymin, xmin, ymax, xmax, keep_topk)
        bb = torch.stack([ymin, xmin, ymax, xmax], dim=-1)
        selected_scores[c], selected_bboxes[c] = nms_bboxes_with_padding(sc, bb, nms_topk, nms_threshold)
    return selected_bboxes, selected_scores
