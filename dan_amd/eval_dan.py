"""Test-time pipeline of the reference's eval_dan.py / eval_sfd.py on the GPU: single-scale detection,
flip test, multi-scale tests and box voting, with the same function names, argument order and return layout
(eval_dan.py:95-297).  Images are uint8 HWC device tensors; detections are device tensors whose rows are
(xmin, ymin, xmax, ymax, score).  The image resize (cv2.resize in the reference) and the voting loop (numpy in the
reference) are libdanhip kernels; everything stays on the device until write_to_txt."""
import numpy as np
import torch

from . import ops
from ._lib import call, lib, ptr, stream

NMS_THRESHOLD = 0.3          # eval_dan.py:69-70
MEMORY_LIMIT = 577.0         # :71-72
MAX_PER_IMAGE = 750          # :73-74
SELECT_THRESHOLD = 0.01      # :75-76


class Detector(object):
    """The `net` argument of the reference's helpers ([sess, image_input, bboxes_pred, cls_pred], eval_dan.py:452):
    image uint8 [H,W,3] -> (bboxes [A,4] as (ymin,xmin,ymax,xmax), scores [A]).  Anchors are cached per image size
    (the reference rebuilds them inside the graph from tf.shape, eval_dan.py:320-342)."""

    def __init__(self, model, anchor_config_fn, precision="act"):
        """precision: "act" = the library build's 16-bit activations (fast path, 3 k img/s); "fp32" = the fp32 inference kernels end to end,
        the path whose box outputs agree with an fp32 reference to 1e-4 (tests/test_eval_f32_gpu.py); "split" = the same graph with every
        convolution as a split-operand product on the fp16 MFMA (csrc/split_infer.hip): the same bound at about three times the fp32 path's rate."""
        if precision not in ("act", "fp32", "split"):
            raise ValueError("precision must be 'act', 'fp32' or 'split'")
        self.model = model
        self.model.precision = precision
        self.anchor_config_fn = anchor_config_fn
        self._anchors = {}

    def anchors(self, h, w, device):
        key = (int(h), int(w))
        if key not in self._anchors:
            self._anchors[key] = self.anchor_config_fn(key[0], key[1], device)
        return self._anchors[key]

    def __call__(self, image):
        assert image.dtype == torch.uint8 and image.dim() == 3 and image.shape[2] == 3
        a = self.anchors(image.shape[0], image.shape[1], image.device)
        boxes, scores = self.model.predict(image.unsqueeze(0).contiguous(), a)
        return boxes[0], scores[0]

    def batch(self, images):
        """B images of one size in ONE forward pass: uint8 [B,H,W,3] -> (bboxes [B,A,4], scores [B,A])."""
        assert images.dtype == torch.uint8 and images.dim() == 4 and images.shape[3] == 3
        a = self.anchors(images.shape[1], images.shape[2], images.device)
        return self.model.predict(images.contiguous(), a)


def resize_image(image, fx, fy):
    """cv2.resize(image, None, None, fx=fx, fy=fy, interpolation=cv2.INTER_LINEAR) (eval_dan.py:97)."""
    image = image.contiguous()
    H, W, C = image.shape
    Wo, Ho = int(np.rint(W * fx)), int(np.rint(H * fy))
    out = torch.empty((Ho, Wo, C), dtype=torch.uint8, device=image.device)
    call("danhip_resize_u8_linear", ptr(image), H, W, ptr(out), Ho, Wo, C, float(fx), float(fy), stream())
    return out


def _order_desc(scores):
    """argsort()[::-1]; ties resolved as a stable ascending sort read backwards (numpy's default sort leaves it open): libdanhip's
    arg-sort with ties_high_index_first.  The detections are fp32 (or fp32 values held in float64: bbox_vote's input); a caller that
    passes scores an fp32 cannot hold gets them rounded to fp32 first."""
    if scores.dtype != torch.float32:
        # every score of this pipeline is an fp32 softmax output (detect_face), at most carried in a float64 container (flip_test,
        # bbox_vote's input): the cast back is exact, and no host round trip checks it
        scores = scores.to(torch.float32)
    return ops.argsort_desc(scores.contiguous(), ties_high_index_first=True)


def detect_face(net, image, shrink, max_per_image=MAX_PER_IMAGE):
    """eval_dan.py:95-118 -> det fp32 [K,5]."""
    if shrink != 1:
        image = resize_image(image, shrink, shrink)
    bboxes, scores = net(image)
    s = torch.tensor(float(shrink), dtype=torch.float32, device=bboxes.device)        # a tensor: true division, not x * (1/s)
    det = torch.stack((bboxes[:, 1] / s, bboxes[:, 0] / s, bboxes[:, 3] / s, bboxes[:, 2] / s, scores.to(torch.float32)), dim=1)
    top = min(det.shape[0] - 1, int(max_per_image * 1.5))
    return det[_order_desc(det[:, 4])[:top]]


def _keep_big(det):
    return det[torch.maximum(det[:, 2] - det[:, 0] + 1, det[:, 3] - det[:, 1] + 1) > 30]


def _keep_small(det):
    return det[torch.minimum(det[:, 2] - det[:, 0] + 1, det[:, 3] - det[:, 1] + 1) < 100]


def multi_scale_test(net, image, max_im_shrink):
    """eval_dan.py:121-148 -> (det_s, det_b)."""
    st = 0.5 if max_im_shrink >= 0.75 else 0.5 * max_im_shrink
    det_s = _keep_big(detect_face(net, image, st))
    bt = min(2, max_im_shrink) if max_im_shrink > 1 else (st + max_im_shrink) / 2
    det_b = detect_face(net, image, bt)
    if max_im_shrink > 2:
        bt *= 2
        while bt < max_im_shrink:
            det_b = torch.cat((det_b, detect_face(net, image, bt)), dim=0)
            bt *= 2
        det_b = torch.cat((det_b, detect_face(net, image, max_im_shrink)), dim=0)
    det_b = _keep_small(det_b) if bt > 1 else _keep_big(det_b)
    return det_s, det_b


def multi_scale_test_pyramid(net, image, max_shrink):
    """eval_dan.py:151-175."""
    det_b = _keep_big(detect_face(net, image, 0.25))
    for st in (0.75, 1.25, 1.5, 1.75):
        if st <= max_shrink:
            d = detect_face(net, image, st)
            d = _keep_small(d) if st > 1 else _keep_big(d)
            det_b = torch.cat((det_b, d), dim=0)
    return det_b


def flip_test(net, image, shrink):
    """eval_dan.py:188-199: detections of the mirrored image mapped back (float32 arithmetic, float64 container)."""
    det_f = detect_face(net, image.flip(1).contiguous(), shrink)
    w = torch.tensor(float(image.shape[1]), dtype=torch.float32, device=det_f.device)
    det_t = torch.empty(det_f.shape, dtype=torch.float64, device=det_f.device)
    det_t[:, 0] = (w - det_f[:, 2]) - 1
    det_t[:, 1] = det_f[:, 1]
    det_t[:, 2] = (w - det_f[:, 0]) - 1
    det_t[:, 3] = det_f[:, 3]
    det_t[:, 4] = det_f[:, 4]
    return det_t


def bbox_vote_batch(dets, nms_threshold=NMS_THRESHOLD, max_per_image=MAX_PER_IMAGE):
    """bbox_vote for a list of per-image detection sets in one launch (one workgroup per image).
    Precondition (as bbox_vote): scores are fp32 values, possibly held in float64 - they are ordered after rounding to fp32."""
    B = len(dets)
    dev = dets[0].device
    nmax = max(1, max(d.shape[0] for d in dets))
    packed = torch.zeros((B, nmax, 5), dtype=torch.float64, device=dev)
    for i, d in enumerate(dets):
        d = d.to(torch.float64)
        packed[i, :d.shape[0]] = d[_order_desc(d[:, 4])]
    counts = torch.tensor([d.shape[0] for d in dets], dtype=torch.int32, device=dev)
    out = torch.empty((B, max_per_image, 5), dtype=torch.float32, device=dev)
    num = torch.empty((B,), dtype=torch.int32, device=dev)
    ws = torch.empty((lib().danhip_bbox_vote_workspace_bytes(B, nmax),), dtype=torch.uint8, device=dev)
    call("danhip_bbox_vote", ptr(packed), ptr(counts), B, nmax, float(nms_threshold), int(max_per_image), ptr(out), ptr(num), ptr(ws), ws.numel(),
         stream())
    n = num.tolist()
    return [out[i, :n[i]] for i in range(B)]


def bbox_vote(det, nms_threshold=NMS_THRESHOLD, max_per_image=MAX_PER_IMAGE):
    """eval_dan.py:201-241 -> fp32 [K,5], K <= max_per_image.
    Precondition: the scores (column 4) are fp32 values, possibly carried in a float64 tensor - what this pipeline produces (softmax outputs
    of detect_face).  They are ordered as fp32 (_order_desc): float64 scores that differ only below fp32 precision become ties and are
    taken higher index first, not in their float64 order."""
    return bbox_vote_batch([det], nms_threshold, max_per_image)[0]


def get_shrink(height, width, memory_limit=MEMORY_LIMIT):
    """eval_dan.py:263-297 -> (shrink, max_shrink); host arithmetic, including the reference's truncation of the
    decimal representation to two places."""
    v1 = (0x7fffffff / memory_limit / (height * width)) ** 0.5
    v2 = ((678 * 1024 * 2.0 * 2.0) / (height * width)) ** 0.5
    x = min(v1, v2)
    text = str(x)
    if '.' not in text:
        raise ValueError("get_shrink: %r has no decimal point (the reference returns None here)" % x)
    head, tail = text.split('.')
    if len(tail) >= 3:
        x = float(head + '.' + tail[:2])
    m = x - 0.3
    if 1.5 <= m < 2:
        m = m - 0.1
    elif 2 <= m < 3:
        m = m - 0.2
    elif 3 <= m < 4:
        m = m - 0.3
    elif 4 <= m < 5:
        m = m - 0.4
    elif m >= 5:
        m = m - 0.5
    return (m if m < 1 else 1), m


def write_to_txt(f, det, event, im_name, select_threshold=SELECT_THRESHOLD):
    """eval_dan.py:243-261 (same text format); det may be a device tensor.  `event` is the WIDER-FACE event entry
    (event[0][0] is its name) or a plain string."""
    det = det.detach().cpu().numpy() if torch.is_tensor(det) else np.asarray(det)
    name = event if isinstance(event, str) else event[0][0]
    xmin, ymin, xmax, ymax, sc = (det[:, i] for i in range(5))
    bh, bw = ymax - ymin + 1, xmax - xmin + 1
    valid = np.logical_and(np.logical_and(np.ceil(bh) >= 10, bw > 1), sc > select_threshold)
    f.write('{:s}\n'.format(name + '/' + im_name + '.jpg'))
    f.write('{}\n'.format(np.count_nonzero(valid)))
    for i in range(valid.shape[0]):
        if valid[i]:
            f.write('{:.1f} {:.1f} {:.1f} {:.1f} {:.3f}\n'.format(np.floor(xmin[i]), np.floor(ymin[i]), np.ceil(bw[i]), np.ceil(bh[i]), sc[i]))


# ---- the same pipeline for B images of ONE size, without a host round trip.  get_shrink and the scale lists of
# multi_scale_test / multi_scale_test_pyramid depend on the image SIZE only, so every pass of eval_dan.py:452-459 runs as one batched
# forward; the size filters (eval_dan.py:126,147-148,153,160-170) become validity masks over fixed-size [B, top] blocks instead of
# boolean compactions (whose output shape the host would have to read), and the voting kernel takes the per-image counts from the device.
def resize_images(images, fx, fy):
    B, H, W, C = images.shape
    Wo, Ho = int(np.rint(W * fx)), int(np.rint(H * fy))
    out = torch.empty((B, Ho, Wo, C), dtype=torch.uint8, device=images.device)
    images = images.contiguous()
    for b in range(B):
        call("danhip_resize_u8_linear", ptr(images[b]), H, W, ptr(out[b]), Ho, Wo, C, float(fx), float(fy), stream())
    return out


def detect_face_batch(net, images, shrink, max_per_image=MAX_PER_IMAGE):
    """detect_face for [B,H,W,3] -> det fp32 [B, top, 5], rows ordered as eval_dan.py:115-116 orders them."""
    if shrink != 1:
        images = resize_images(images, shrink, shrink)
    bboxes, scores = net.batch(images)
    s = torch.tensor(float(shrink), dtype=torch.float32, device=bboxes.device)
    det = torch.stack((bboxes[..., 1] / s, bboxes[..., 0] / s, bboxes[..., 3] / s, bboxes[..., 2] / s, scores.to(torch.float32)), dim=2)
    top = min(det.shape[1] - 1, int(max_per_image * 1.5))
    order = torch.stack([_order_desc(det[b, :, 4])[:top] for b in range(det.shape[0])])
    return torch.gather(det, 1, order.unsqueeze(2).expand(-1, -1, 5))


def _big_mask(det):
    return torch.maximum(det[..., 2] - det[..., 0] + 1, det[..., 3] - det[..., 1] + 1) > 30


def _small_mask(det):
    return torch.minimum(det[..., 2] - det[..., 0] + 1, det[..., 3] - det[..., 1] + 1) < 100


def detect_images(net, images, pyramid=True, nms_threshold=NMS_THRESHOLD, max_per_image=MAX_PER_IMAGE):
    """detect_image for a batch: uint8 [B,H,W,3] -> (dets fp32 [B, max_per_image, 5], num int32 [B]) on the device; rows beyond num[b]
    are unspecified.  Per image the result equals detect_image(net, images[b], pyramid) whenever the forward passes agree (a batched
    forward may take another split-K plan than a single image: same math, another fp32 summation order)."""
    B, H, W, _ = images.shape
    shrink, max_shrink = get_shrink(H, W)
    parts = []                                             # (det [B, k, 5] float64, valid [B, k])
    def add(det, mask=None):
        parts.append((det.to(torch.float64), torch.ones(det.shape[:2], dtype=torch.bool, device=det.device) if mask is None else mask))
    add(detect_face_batch(net, images, shrink))
    # flip_test (eval_dan.py:188-199)
    det_f = detect_face_batch(net, images.flip(2).contiguous(), shrink)
    w = torch.tensor(float(W), dtype=torch.float32, device=det_f.device)
    det_t = torch.empty(det_f.shape, dtype=torch.float64, device=det_f.device)
    det_t[..., 0] = (w - det_f[..., 2]) - 1
    det_t[..., 1] = det_f[..., 1]
    det_t[..., 2] = (w - det_f[..., 0]) - 1
    det_t[..., 3] = det_f[..., 3]
    det_t[..., 4] = det_f[..., 4]
    add(det_t)
    # multi_scale_test (eval_dan.py:121-148)
    st = 0.5 if max_shrink >= 0.75 else 0.5 * max_shrink
    d = detect_face_batch(net, images, st)
    add(d, _big_mask(d))
    bt = min(2, max_shrink) if max_shrink > 1 else (st + max_shrink) / 2
    bs = [detect_face_batch(net, images, bt)]
    if max_shrink > 2:
        bt *= 2
        while bt < max_shrink:
            bs.append(detect_face_batch(net, images, bt))
            bt *= 2
        bs.append(detect_face_batch(net, images, max_shrink))
    for d in bs:
        add(d, _small_mask(d) if bt > 1 else _big_mask(d))
    # multi_scale_test_pyramid (eval_dan.py:151-175)
    if pyramid:
        d = detect_face_batch(net, images, 0.25)
        add(d, _big_mask(d))
        for sc in (0.75, 1.25, 1.5, 1.75):
            if sc <= max_shrink:
                d = detect_face_batch(net, images, sc)
                add(d, _small_mask(d) if sc > 1 else _big_mask(d))
    det = torch.cat([p[0] for p in parts], dim=1)
    valid = torch.cat([p[1] for p in parts], dim=1)
    n = det.shape[1]
    # voting input: the valid rows in descending score order (ties: higher index first, as np.argsort()[::-1]); invalid rows carry -inf
    # and fall behind every real score (scores are softmax outputs >= 0), and the relative index order of the valid rows is the order of
    # the serial concatenation, so ties resolve alike
    key = torch.where(valid, det[..., 4].to(torch.float32), torch.full((), float("-inf"), dtype=torch.float32, device=det.device))
    order = torch.stack([ops.argsort_desc(key[b].contiguous(), ties_high_index_first=True) for b in range(B)])
    packed = torch.gather(det, 1, order.unsqueeze(2).expand(-1, -1, 5)).contiguous()
    counts = valid.sum(dim=1).to(torch.int32)
    out = torch.empty((B, max_per_image, 5), dtype=torch.float32, device=det.device)
    num = torch.empty((B,), dtype=torch.int32, device=det.device)
    ws = torch.empty((lib().danhip_bbox_vote_workspace_bytes(B, n),), dtype=torch.uint8, device=det.device)
    call("danhip_bbox_vote", ptr(packed), ptr(counts), B, n, float(nms_threshold), int(max_per_image), ptr(out), ptr(num), ptr(ws), ws.numel(), stream())
    return out, num


def detect_image(net, image, pyramid=True):
    """The per-image body of eval_dan.py:452-459 (eval_sfd.py:323-328 without the pyramid pass)."""
    shrink, max_shrink = get_shrink(image.shape[0], image.shape[1])
    dets = [detect_face(net, image, shrink), flip_test(net, image, shrink)]
    dets += list(multi_scale_test(net, image, max_shrink))
    if pyramid:
        dets.append(multi_scale_test_pyramid(net, image, max_shrink))
    return bbox_vote(torch.cat([d.to(torch.float64) for d in dets], dim=0))
