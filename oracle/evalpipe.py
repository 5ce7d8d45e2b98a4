"""TEST INFRASTRUCTURE ONLY — numpy restatement of the reference's test-time pipeline (eval_dan.py:95-297, eval_sfd.py:60-196).

`net` below is any callable image_u8[H,W,3] -> (bboxes float32 [A,4] as (ymin,xmin,ymax,xmax), scores float32 [A]) — the
`sess.run(net[2:], {net[1]: image})` of eval_dan.py:99.

Pinning: the reference holds no fixtures for this code and cv2 / TensorFlow are absent here, so cv2.resize is restated from
OpenCV's published generic INTER_LINEAR 8-bit algorithm (imgproc/resize.cpp: 11-bit coefficients, HResizeLinear then
VResizeLinear with FixedPtCast<int,uchar,22>) — **parity unpinned** against a real cv2 build (builds with IPP take another
path).  bbox_vote / get_shrink / write_to_txt are numpy / pure-python in the reference and are restated statement by
statement; tie order of `argsort()[::-1]` (numpy's default sort is not stable) is pinned to "stable ascending, reversed"."""
import numpy as np

NMS_THRESHOLD = 0.3          # eval_dan.py:69-70
MEMORY_LIMIT = 577.0         # :71-72
MAX_PER_IMAGE = 750          # :73-74
SELECT_THRESHOLD = 0.01      # :75-76


def _round_short(v):
    return np.clip(np.rint(v), -32768, 32767).astype(np.int32)


def cv2_resize_linear_u8(img, fx, fy):
    """cv2.resize(img, None, None, fx=fx, fy=fy, interpolation=cv2.INTER_LINEAR) for uint8 HWC (eval_dan.py:97)."""
    H, W, C = img.shape
    Wo, Ho = int(np.rint(W * fx)), int(np.rint(H * fy))
    sx_scale, sy_scale = 1.0 / fx, 1.0 / fy
    dx = np.arange(Wo, dtype=np.float64)
    fxs = ((dx + 0.5) * sx_scale - 0.5).astype(np.float32)
    sx = np.floor(fxs).astype(np.int64)
    fxs = fxs - sx.astype(np.float32)
    lo = sx < 0
    fxs[lo] = 0; sx[lo] = 0
    edge = sx >= W - 1
    fxs[edge] = 0; sx[edge] = W - 1
    a0 = _round_short((np.float32(1) - fxs) * np.float32(2048))
    a1 = _round_short(fxs * np.float32(2048))
    sx1 = np.where(edge, sx, sx + 1)
    dy = np.arange(Ho, dtype=np.float64)
    fys = ((dy + 0.5) * sy_scale - 0.5).astype(np.float32)
    sy = np.floor(fys).astype(np.int64)
    fys = fys - sy.astype(np.float32)
    b0 = _round_short((np.float32(1) - fys) * np.float32(2048))
    b1 = _round_short(fys * np.float32(2048))
    sy0 = np.clip(sy, 0, H - 1)
    sy1 = np.clip(sy + 1, 0, H - 1)
    src = img.astype(np.int32)
    # horizontal pass on the needed rows
    def hpass(rows):
        r = src[rows]                                            # [Ho, W, C]
        h = r[:, sx, :] * a0[None, :, None] + r[:, sx1, :] * a1[None, :, None]
        he = r[:, sx, :] * 2048
        return np.where(edge[None, :, None], he, h)
    r0, r1 = hpass(sy0), hpass(sy1)
    v = (((b0[:, None, None] * (r0 >> 4)) >> 16) + ((b1[:, None, None] * (r1 >> 4)) >> 16) + 2) >> 2
    return np.clip(v, 0, 255).astype(np.uint8)


def _order_desc(scores):
    """argsort()[::-1] with the tie order pinned (stable ascending, reversed)."""
    return np.argsort(scores, kind="stable")[::-1]


def detect_face(net, image, shrink, max_per_image=MAX_PER_IMAGE):
    """eval_dan.py:95-118."""
    if shrink != 1:
        image = cv2_resize_linear_u8(image, shrink, shrink)
    bboxes, scores = net(image)
    s = np.float32(shrink)
    det = np.column_stack((bboxes[:, 1] / s, bboxes[:, 0] / s, bboxes[:, 3] / s, bboxes[:, 2] / s, scores))
    top = min(det.shape[0] - 1, int(max_per_image * 1.5))
    return det[_order_desc(det[:, 4])[:top], :]


def _keep_big(det):
    return det[np.maximum(det[:, 2] - det[:, 0] + 1, det[:, 3] - det[:, 1] + 1) > 30]


def _keep_small(det):
    return det[np.minimum(det[:, 2] - det[:, 0] + 1, det[:, 3] - det[:, 1] + 1) < 100]


def multi_scale_test(net, image, max_im_shrink):
    """eval_dan.py:121-148."""
    st = 0.5 if max_im_shrink >= 0.75 else 0.5 * max_im_shrink
    det_s = _keep_big(detect_face(net, image, st))
    bt = min(2, max_im_shrink) if max_im_shrink > 1 else (st + max_im_shrink) / 2
    det_b = detect_face(net, image, bt)
    if max_im_shrink > 2:
        bt *= 2
        while bt < max_im_shrink:
            det_b = np.vstack((det_b, detect_face(net, image, bt)))
            bt *= 2
        det_b = np.vstack((det_b, detect_face(net, image, max_im_shrink)))
    det_b = _keep_small(det_b) if bt > 1 else _keep_big(det_b)
    return det_s, det_b


def multi_scale_test_pyramid(net, image, max_shrink):
    """eval_dan.py:151-175."""
    det_b = _keep_big(detect_face(net, image, 0.25))
    for st in (0.75, 1.25, 1.5, 1.75):
        if st <= max_shrink:
            d = detect_face(net, image, st)
            d = _keep_small(d) if st > 1 else _keep_big(d)
            det_b = np.vstack((det_b, d))
    return det_b


def flip_test(net, image, shrink):
    """eval_dan.py:188-199 (result array is float64, the arithmetic float32)."""
    det_f = detect_face(net, image[:, ::-1, :], shrink)
    det_t = np.zeros(det_f.shape)
    det_t[:, 0] = image.shape[1] - det_f[:, 2] - 1
    det_t[:, 1] = det_f[:, 1]
    det_t[:, 2] = image.shape[1] - det_f[:, 0] - 1
    det_t[:, 3] = det_f[:, 3]
    det_t[:, 4] = det_f[:, 4]
    return det_t


def bbox_vote(det, nms_threshold=NMS_THRESHOLD, max_per_image=MAX_PER_IMAGE):
    """eval_dan.py:201-241 with an alive mask instead of np.delete (same visiting order, same float64 arithmetic)."""
    det = np.asarray(det)
    det = det[_order_desc(det[:, 4])]
    n = det.shape[0]
    alive = np.ones(n, dtype=bool)
    area = (det[:, 2] - det[:, 0] + 1) * (det[:, 3] - det[:, 1] + 1)
    out = []
    head = 0
    while True:
        while head < n and not alive[head]:
            head += 1
        if head >= n:
            break
        idx = np.nonzero(alive)[0]
        d = det[idx]
        w = np.maximum(0.0, np.minimum(det[head, 2], d[:, 2]) - np.maximum(det[head, 0], d[:, 0]) + 1)
        h = np.maximum(0.0, np.minimum(det[head, 3], d[:, 3]) - np.maximum(det[head, 1], d[:, 1]) + 1)
        inter = w * h
        with np.errstate(invalid="ignore", divide="ignore"):
            o = inter / (area[head] + area[idx] - inter)
        merge = idx[o >= nms_threshold]
        alive[merge] = False
        if merge.shape[0] == 0:
            alive[head] = False
        if merge.shape[0] <= 1:
            continue
        acc = det[merge].copy()
        acc[:, 0:4] = acc[:, 0:4] * acc[:, 4:5]
        row = np.zeros((1, 5), dtype=np.float32)
        row[:, 0:4] = np.sum(acc[:, 0:4], axis=0) / np.sum(acc[:, 4:5])
        row[:, 4] = np.max(acc[:, 4])
        out.append(row)
    dets = np.vstack(out) if out else np.zeros((0, 5), dtype=np.float32)
    return dets[:max_per_image].astype(np.float32)


def get_shrink(height, width, memory_limit=MEMORY_LIMIT):
    """eval_dan.py:263-297 (the string-based truncation included: values whose repr has < 3 decimals pass through,
    values without a '.' in their repr yield None in the reference — not reachable for image-sized inputs)."""
    v1 = (0x7fffffff / memory_limit / (height * width)) ** 0.5
    v2 = ((678 * 1024 * 2.0 * 2.0) / (height * width)) ** 0.5
    x = min(v1, v2)
    s = str(x)
    before, after = s.split('.')
    x = float(before + '.' + after[0:2]) if len(after) >= 3 else x
    m = x - 0.3
    if 1.5 <= m < 2:
        m -= 0.1
    elif 2 <= m < 3:
        m -= 0.2
    elif 3 <= m < 4:
        m -= 0.3
    elif 4 <= m < 5:
        m -= 0.4
    elif m >= 5:
        m -= 0.5
    return (m if m < 1 else 1), m


def format_detections(det, header, select_threshold=SELECT_THRESHOLD):
    """write_to_txt of eval_dan.py:243-261 -> list of lines (header = '<event>/<name>.jpg')."""
    xmin, ymin, xmax, ymax, sc = (det[:, i] for i in range(5))
    bh, bw = ymax - ymin + 1, xmax - xmin + 1
    valid = (np.ceil(bh) >= 10) & (bw > 1) & (sc > select_threshold)
    lines = [header, str(int(np.count_nonzero(valid)))]
    for i in np.nonzero(valid)[0]:
        lines.append('{:.1f} {:.1f} {:.1f} {:.1f} {:.3f}'.format(np.floor(xmin[i]), np.floor(ymin[i]), np.ceil(bw[i]), np.ceil(bh[i]), sc[i]))
    return lines


def detect_image(net, image, pyramid=True):
    """Loop body of eval_dan.py:452-459 (eval_sfd.py has no pyramid pass)."""
    shrink, max_shrink = get_shrink(image.shape[0], image.shape[1])
    dets = [detect_face(net, image, shrink), flip_test(net, image, shrink)]
    dets += list(multi_scale_test(net, image, max_shrink))
    if pyramid:
        dets.append(multi_scale_test_pyramid(net, image, max_shrink))
    return bbox_vote(np.vstack(dets))
