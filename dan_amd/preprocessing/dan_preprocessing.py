"""Training input pipeline of the reference's preprocessing/dan_preprocessing.py on the GPU.

Same function names and meaning as the reference (distort_color, dan_random_sample_patch_wrapper,
pyramid_box_random_sample_patch_wrapper, dan_random_sample, data_anchor_sampling, random_flip_left_right, preprocess_for_train),
re-cut for the device: the geometric functions decide a crop WINDOW and transform the boxes (host arithmetic on a handful of
boxes, float32 like the reference's tf ops), the colour function decides the op chain, and one libdanhip launch
(danhip_augment_preprocess) produces the network input from the decoded uint8 image — nothing is materialised in between.
Random numbers come from a `draws` object with uniform(lo, hi) / randint(lo, hi) / choice(n) (default: numpy RandomState; the
reference's TF stream is not reproducible), so a fixed seed gives a fixed augmentation."""
import ctypes

import numpy as np
import torch

from .._lib import ACT_DTYPE, call, lib, ptr, stream

F = np.float32
OP_CODES = {"brightness": 0, "saturation": 1, "hue": 2, "contrast": 3}
_ORDERINGS = {0: ("brightness", "saturation", "hue", "contrast"), 1: ("saturation", "brightness", "contrast", "hue"),
              2: ("contrast", "hue", "brightness", "saturation"), 3: ("hue", "saturation", "contrast", "brightness")}     # :117-145


class Draws(object):
    def __init__(self, seed=None):
        self.r = np.random.RandomState(seed)

    def uniform(self, lo, hi):
        return F(F(lo) + F(self.r.random_sample()) * (F(hi) - F(lo)))

    def randint(self, lo, hi):
        return int(lo) if hi <= lo else int(self.r.randint(int(lo), int(hi)))

    def choice(self, n):
        return int(self.r.randint(0, n))


def distort_color(color_ordering, draws, fast_mode=False):
    """dan_preprocessing.py:98-150 -> the op chain [(name, value)] (applied, then clipped to [0,1], by the device kernel)."""
    if color_ordering not in _ORDERINGS:
        raise ValueError("color_ordering must be in [0, 3]")
    names = _ORDERINGS[color_ordering]
    if fast_mode:
        names = ("brightness", "saturation") if color_ordering == 0 else ("saturation", "brightness")
    ops = []
    for n in names:
        if n == "brightness":
            ops.append((n, draws.uniform(-32. / 255., 32. / 255.)))
        elif n == "hue":
            ops.append((n, draws.uniform(-0.2, 0.2)))
        else:
            ops.append((n, draws.uniform(0.5, 1.5)))
    return ops


def _clip_to_window(b, h, w):
    ymin, xmin = np.maximum(F(0), b[:, 0]), np.maximum(F(0), b[:, 1])
    ymax, xmax = np.minimum(F(h) - F(1), b[:, 2]), np.minimum(F(w) - F(1), b[:, 3])
    return np.stack([np.minimum(ymin, ymax), np.minimum(xmin, xmax), ymax, xmax], -1).astype(F)


def dan_random_sample_patch_wrapper(height, width, bboxes, draws):
    """dan_preprocessing.py:410-493: a square window of 0.3 .. 1 x the shorter side that contains a box centre (25 attempts, then a
    window around a random box) -> ((y, x, h, w), boxes in window coordinates)."""
    fh, fw = F(height), F(width)
    patch_list = [draws.uniform(0.3, 1.) for _ in range(4)] + [F(1.)]
    side = int(F(patch_list[draws.choice(5)]) * min(fh, fw))
    cy, cx = (bboxes[:, 0] + bboxes[:, 2]) / F(2), (bboxes[:, 1] + bboxes[:, 3]) / F(2)
    index, mask, roi = 0, np.zeros(len(bboxes), bool), None
    while (mask.sum() < 1 and index < 25) or index < 1:
        x = draws.randint(0, width - side + 1)
        y = draws.randint(0, height - side + 1)
        roi = [F(y), F(x), F(y + side) - F(1), F(x + side) - F(1)]
        mask = (cy > roi[0]) & (cx > roi[1]) & (cy < roi[2]) & (cx < roi[3])
        index += 1
    if mask.sum() == 0:                                                        # sample_around_bbox
        t = draws.randint(0, len(bboxes))
        rcx, rcy = (bboxes[t, 1] + bboxes[t, 3]) / F(2), (bboxes[t, 0] + bboxes[t, 2]) / F(2)
        half = F(side) / F(2)
        roi = [max(rcy - half, F(0)), max(rcx - half, F(0)), min(rcy + half, fh - F(1)), min(rcx + half, fw - F(1))]
        mask = (cy >= roi[0]) & (cx >= roi[1]) & (cy <= roi[2]) & (cx <= roi[3])
    win = (int(roi[0]), int(roi[1]), int(roi[2] - roi[0] + F(1)), int(roi[3] - roi[1] + F(1)))
    b = (bboxes[mask] - np.asarray([win[0], win[1], win[0], win[1]], F)).astype(F)
    return win, _clip_to_window(b, win[2], win[3])


def pyramid_box_random_sample_patch_wrapper(height, width, bboxes, select_face_ind, patch_size, draws):
    """dan_preprocessing.py:495-565: a patch_size square placed so that it covers the selected face; it may leave the image (the
    outside is filled with the mean colour by the kernel) -> ((y, x, s, s), boxes of the faces whose centre is inside)."""
    ps = int(patch_size)
    f = bboxes[select_face_ind]
    fy0, fx0 = max(int(np.floor(f[0])), 0), max(int(np.floor(f[1])), 0)
    fy1, fx1 = min(int(np.ceil(f[2])), height - 1), min(int(np.ceil(f[3])), width - 1)
    fcx, fcy = int(np.floor(F(fx0 + fx1) / F(2))), int(np.floor(F(fy0 + fy1) / F(2)))
    lo, hi = sorted((min(fx1 - ps + 1, fcx), fx0))
    xmin = draws.randint(lo, hi + 1)
    lo, hi = sorted((min(fy1 - ps + 1, fcy), fy0))
    ymin = draws.randint(lo, hi + 1)
    pad_l, pad_t = max(-xmin, 0), max(-ymin, 0)
    b = (bboxes + np.asarray([pad_t, pad_l, pad_t, pad_l], F)).astype(F)       # boxes in padded-image coordinates
    X0, Y0 = xmin + pad_l, ymin + pad_t
    X1, Y1 = X0 + ps - 1, Y0 + ps - 1
    cx, cy = (b[:, 1] + b[:, 3]) / F(2), (b[:, 0] + b[:, 2]) / F(2)
    b = b[(cy > F(Y0)) & (cx > F(X0)) & (cy < F(Y1)) & (cx < F(X1))]
    ymin_c, xmin_c = np.maximum(F(0), b[:, 0] - F(Y0)), np.maximum(F(0), b[:, 1] - F(X0))
    ymax_c, xmax_c = np.minimum(F(Y1), b[:, 2]) - F(Y0), np.minimum(F(X1), b[:, 3]) - F(X0)
    boxes = np.stack([np.minimum(ymin_c, ymax_c), np.minimum(xmin_c, xmax_c), ymax_c, xmax_c], -1).astype(F)
    return (ymin, xmin, ps, ps), boxes


def dan_random_sample(height, width, bboxes, out_shape, draws):
    """dan_preprocessing.py:623-635."""
    win, b = dan_random_sample_patch_wrapper(height, width, bboxes, draws)
    th, tw, fh, fw = F(out_shape[0]), F(out_shape[1]), F(win[2]), F(win[3])
    return win, np.stack([b[:, 0] * th / fh, b[:, 1] * tw / fw, b[:, 2] * th / fh, b[:, 3] * tw / fw], -1).astype(F)


def data_anchor_sampling(height, width, bboxes, anchor_scales, out_shape, draws):
    """dan_preprocessing.py:637-675 (PyramidBox data-anchor-sampling: resize a random face towards a random smaller-or-equal anchor)."""
    assert out_shape[0] == out_shape[1], "output patch should be square!"
    fh_, fw_ = bboxes[:, 2] - bboxes[:, 0], bboxes[:, 3] - bboxes[:, 1]
    face_scale = np.maximum(fw_, fh_)
    sel = draws.randint(0, len(face_scale))
    sel_scale = max(face_scale[sel], F(16.))
    scales = np.asarray(anchor_scales, F)
    anchor_ind = min(int(np.argmax(-np.abs(scales - sel_scale))) + 1, len(scales) - 1) + 1
    target = scales[draws.randint(0, anchor_ind)]
    final_scale = draws.uniform(target / F(2.), min(F(2.) * np.sqrt(fh_[sel] * fw_[sel]), target * F(2.))) / sel_scale
    patch_size = min(max(F(out_shape[0]) / final_scale, F(64.)), min(F(height), F(width)) * F(8.))
    win, b = pyramid_box_random_sample_patch_wrapper(height, width, bboxes, sel, patch_size, draws)
    return win, (b * (F(out_shape[0]) / F(win[2]))).astype(F)


def random_flip_left_right(bboxes, width, draws):
    """dan_preprocessing.py:609-621 -> (flip flag, boxes)."""
    flip = bool(draws.uniform(0., 1.) < 0.5)
    if flip:
        w = F(width)
        bboxes = np.stack([bboxes[:, 0], w - F(1.) - bboxes[:, 3], bboxes[:, 2], w - F(1.) - bboxes[:, 1]], -1).astype(F)
    return flip, bboxes


def augment_image(image_u8, ops, window, flip, out_shape):
    """The device pass: uint8 [H,W,3] device tensor -> network input [out_h,out_w,8] (16-bit, BGR, mean-subtracted, zero padded)."""
    assert image_u8.dtype == torch.uint8 and image_u8.dim() == 3 and image_u8.shape[2] == 3 and image_u8.is_contiguous()
    H, W = int(image_u8.shape[0]), int(image_u8.shape[1])
    out = torch.empty((int(out_shape[0]), int(out_shape[1]), 8), dtype=ACT_DTYPE, device=image_u8.device)
    codes = (ctypes.c_int32 * 4)(*([OP_CODES[n] for n, _ in ops] + [0] * (4 - len(ops))))
    vals = (ctypes.c_float * 4)(*([float(v) for _, v in ops] + [0.0] * (4 - len(ops))))
    nws = lib().danhip_augment_workspace_bytes()
    ws = torch.empty(nws, dtype=torch.uint8, device=image_u8.device)
    call("danhip_augment_preprocess", ptr(image_u8), H, W, len(ops), codes, vals, int(window[0]), int(window[1]), int(window[2]), int(window[3]),
         int(bool(flip)), ptr(out), int(out_shape[0]), int(out_shape[1]), ptr(ws), nws, stream())
    out._real_channels = 3
    return out


def preprocess_for_train(image, bboxes, out_shape, anchor_scales, data_format="channels_last", scope=None, output_rgb=False, draws=None):
    """dan_preprocessing.py:677-733.  image: uint8 [H,W,3] RGB device tensor; bboxes: float32 [n,4] (ymin,xmin,ymax,xmax) in pixels.
    -> (network input [out_h,out_w,8] 16-bit BGR mean-subtracted, boxes float32 [k,4] numpy in output pixels; small boxes dropped)."""
    if data_format != "channels_last" or output_rgb:
        raise ValueError("the MI355X build feeds NHWC BGR (the reference's training scripts pass output_rgb=False)")
    draws = draws or Draws()
    bboxes = np.asarray(bboxes.detach().cpu().numpy() if torch.is_tensor(bboxes) else bboxes, F)
    H, W = int(image.shape[0]), int(image.shape[1])
    ops = distort_color(draws.randint(0, 4), draws, fast_mode=False)                 # apply_with_random_selector(.., num_cases=4)
    if draws.uniform(0., 1.) < 0.5:
        window, boxes = dan_random_sample(H, W, bboxes, out_shape, draws)
    else:
        window, boxes = data_anchor_sampling(H, W, bboxes, anchor_scales, out_shape, draws)
    flip, boxes = random_flip_left_right(boxes, out_shape[1], draws)
    keep = ((boxes[:, 2] - boxes[:, 0]) > F(6.)) & ((boxes[:, 3] - boxes[:, 1]) > F(3.))
    return augment_image(image, ops, window, flip, out_shape), boxes[keep]
