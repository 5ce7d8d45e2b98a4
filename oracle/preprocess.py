"""TEST INFRASTRUCTURE ONLY — numpy float32 restatement of the reference's training input pipeline
(preprocessing/dan_preprocessing.py:98-150 distort_color, :410-493 dan_random_sample_patch_wrapper, :495-565
pyramid_box_random_sample_patch_wrapper, :609-675 flip / dan_random_sample / data_anchor_sampling, :677-733 preprocess_for_train).

Every image op is materialised on the full image exactly in the reference's order (distort colours -> crop / pad -> legacy bilinear
resize -> flip -> uint8 conversion -> mean subtraction -> BGR).  The tf.image kernels it calls (adjust_saturation / adjust_hue via
HSV, adjust_contrast about the per-channel mean, resize_bilinear with align_corners=False, convert_image_dtype) are TensorFlow code
that is not in the reference tree: restated from their published formulas — **parity unpinned** against TF.  TF's random stream cannot
be reproduced, so every random draw is taken from a caller-supplied `Draws` object (same interface on the product side): parity is
"same draws -> same image and boxes"; the distribution of the draws follows the reference statement by statement."""
import numpy as np

F = np.float32
MEANS_RGB = (F(123.68), F(116.78), F(103.94))


class Draws(object):
    """The random primitives the reference uses, on a numpy RandomState (tf.random_uniform float / int, tf.multinomial uniform)."""

    def __init__(self, seed):
        self.r = np.random.RandomState(seed)

    def uniform(self, lo, hi):
        return F(F(lo) + F(self.r.random_sample()) * (F(hi) - F(lo)))

    def randint(self, lo, hi):                       # [lo, hi)
        return int(lo) if hi <= lo else int(self.r.randint(int(lo), int(hi)))

    def choice(self, n):
        return int(self.r.randint(0, n))


# ------------------------------------------------------------------------------------------------ tf.image colour ops (float32)
def rgb_to_hsv(r, g, b):
    M = np.maximum(np.maximum(r, g), b)
    m = np.minimum(np.minimum(r, g), b)
    c = (M - m).astype(F)
    safe = np.where(c > 0, c, F(1))
    hr = ((g - b) / safe).astype(F)
    hr = np.where(hr < 0, hr + F(6), hr)
    hg = ((b - r) / safe + F(2)).astype(F)
    hb = ((r - g) / safe + F(4)).astype(F)
    h = np.where(M == r, hr, np.where(M == g, hg, hb)) / F(6)
    h = np.where(c > 0, h, F(0)).astype(F)
    s = np.where(M > 0, c / np.where(M > 0, M, F(1)), F(0)).astype(F)
    return h, s, M.astype(F)


def hsv_to_rgb(h, s, v):
    c = (s * v).astype(F)
    m = (v - c).astype(F)
    dh = (h * F(6)).astype(F)
    fm = np.fmod(dh, F(2)).astype(F)
    x = (c * (F(1) - np.abs(fm - F(1)))).astype(F)
    k = np.minimum(dh.astype(np.int32), 5)
    z = np.zeros_like(c)
    r = np.choose(k, [c, x, z, z, x, c])
    g = np.choose(k, [x, c, c, x, z, z])
    b = np.choose(k, [z, z, x, c, c, x])
    return (r + m).astype(F), (g + m).astype(F), (b + m).astype(F)


def adjust_saturation(img, factor):
    h, s, v = rgb_to_hsv(img[..., 0], img[..., 1], img[..., 2])
    s = np.clip(s * F(factor), F(0), F(1)).astype(F)
    return np.stack(hsv_to_rgb(h, s, v), -1)


def adjust_hue(img, delta):
    h, s, v = rgb_to_hsv(img[..., 0], img[..., 1], img[..., 2])
    h = (h + F(delta)).astype(F)
    h = (h - np.floor(h)).astype(F)                    # wrap into [0, 1)
    return np.stack(hsv_to_rgb(h, s, v), -1)


def adjust_contrast(img, factor, mean=None):
    mean = img.reshape(-1, 3).mean(0, dtype=np.float64).astype(F) if mean is None else np.asarray(mean, F)
    return ((img - mean) * F(factor) + mean).astype(F)


ORDERINGS = {0: ("brightness", "saturation", "hue", "contrast"), 1: ("saturation", "brightness", "contrast", "hue"),
             2: ("contrast", "hue", "brightness", "saturation"), 3: ("hue", "saturation", "contrast", "brightness")}


def draw_color_params(d):
    """apply_with_random_selector(.., num_cases=4) + the four tf.image.random_* draws of distort_color(fast_mode=False) (:117-145),
    drawn in the order the chosen ordering applies them."""
    ordering = d.randint(0, 4)
    ops = []
    for name in ORDERINGS[ordering]:
        if name == "brightness":
            ops.append((name, d.uniform(-32. / 255., 32. / 255.)))
        elif name == "hue":
            ops.append((name, d.uniform(-0.2, 0.2)))
        else:
            ops.append((name, d.uniform(0.5, 1.5)))
    return ops


def distort_color(img01, ops):
    """img01 float32 [H,W,3] in [0,1]; ops = [(name, value)] -> (distorted image clipped to [0,1], contrast mean or None)."""
    mean = None
    for name, val in ops:
        if name == "brightness":
            img01 = (img01 + F(val)).astype(F)
        elif name == "saturation":
            img01 = adjust_saturation(img01, val)
        elif name == "hue":
            img01 = adjust_hue(img01, val)
        else:
            mean = img01.reshape(-1, 3).mean(0, dtype=np.float64).astype(F)
            img01 = adjust_contrast(img01, val, mean)
    return np.clip(img01, F(0), F(1)).astype(F), mean


# ------------------------------------------------------------------------------------------------ patch sampling (host logic)
def dan_random_sample_window(height, width, bboxes, d):
    """dan_random_sample_patch_wrapper (:410-493) -> ((y, x, h, w) int window inside the image, transformed boxes float32 [k,4])."""
    fh, fw = F(height), F(width)
    patches = [d.uniform(0.3, 1.) for _ in range(4)] + [F(1.)]
    side = int(F(patches[d.choice(5)]) * min(fh, fw))                       # tf.to_int32 truncates
    cy, cx = (bboxes[:, 0] + bboxes[:, 2]) / F(2), (bboxes[:, 1] + bboxes[:, 3]) / F(2)
    index, mask, roi = 0, np.zeros(len(bboxes), bool), [F(0), F(0), fh - 1, fw - 1]
    while (mask.sum() < 1 and index < 25) or index < 1:
        x = d.randint(0, width - side + 1)
        y = d.randint(0, height - side + 1)
        roi = [F(y), F(x), F(y + side) - F(1), F(x + side) - F(1)]
        mask = (cy > roi[0]) & (cx > roi[1]) & (cy < roi[2]) & (cx < roi[3])
        index += 1
    if mask.sum() > 0:
        win = [int(roi[0]), int(roi[1]), int(roi[2] - roi[0] + F(1)), int(roi[3] - roi[1] + F(1))]
        kept = bboxes[mask]
    else:                                                                    # sample_around_bbox (:414-431)
        t = d.randint(0, len(bboxes))
        rcx, rcy = (bboxes[t, 1] + bboxes[t, 3]) / F(2), (bboxes[t, 0] + bboxes[t, 2]) / F(2)
        half = F(side) / F(2)
        roi = [max(rcy - half, F(0)), max(rcx - half, F(0)), min(rcy + half, fh - F(1)), min(rcx + half, fw - F(1))]
        m2 = (cy >= roi[0]) & (cx >= roi[1]) & (cy <= roi[2]) & (cx <= roi[3])
        win = [int(roi[0]), int(roi[1]), int(roi[2] - roi[0] + F(1)), int(roi[3] - roi[1] + F(1))]
        kept = bboxes[m2]
    off = np.asarray([win[0], win[1], win[0], win[1]], F)
    b = (kept - off).astype(F)
    ymin, xmin = np.maximum(F(0), b[:, 0]), np.maximum(F(0), b[:, 1])
    ymax, xmax = np.minimum(F(win[2]) - F(1), b[:, 2]), np.minimum(F(win[3]) - F(1), b[:, 3])
    ymin, xmin = np.minimum(ymin, ymax), np.minimum(xmin, xmax)
    return tuple(win), np.stack([ymin, xmin, ymax, xmax], -1).astype(F)


def anchor_sample_window(height, width, bboxes, select_ind, patch_size, d):
    """pyramid_box_random_sample_patch_wrapper (:495-565) -> ((y, x, size, size) window that may leave the image: the outside is
    filled with the mean colour; boxes in window coordinates)."""
    ps = int(patch_size)
    f = bboxes[select_ind]
    fy0, fx0 = max(int(np.floor(f[0])), 0), max(int(np.floor(f[1])), 0)
    fy1, fx1 = min(int(np.ceil(f[2])), height - 1), min(int(np.ceil(f[3])), width - 1)
    fcx, fcy = int(np.floor(F(fx0 + fx1) / F(2))), int(np.floor(F(fy0 + fy1) / F(2)))
    xl, xh = min(min(fx1 - ps + 1, fcx), fx0), max(min(fx1 - ps + 1, fcx), fx0)
    xmin = d.randint(xl, xh + 1)
    yl, yh = min(min(fy1 - ps + 1, fcy), fy0), max(min(fy1 - ps + 1, fcy), fy0)
    ymin = d.randint(yl, yh + 1)
    xmax, ymax = xmin + ps - 1, ymin + ps - 1
    pad_l, pad_t = (-xmin if xmin < 0 else 0), (-ymin if ymin < 0 else 0)
    b = (bboxes + np.asarray([pad_t, pad_l, pad_t, pad_l], F)).astype(F)
    X0, Y0, X1, Y1 = xmin + pad_l, ymin + pad_t, xmax + pad_l, ymax + pad_t          # window in padded-image coordinates
    cx, cy = (b[:, 1] + b[:, 3]) / F(2), (b[:, 0] + b[:, 2]) / F(2)
    keep = (cy > F(Y0)) & (cx > F(X0)) & (cy < F(Y1)) & (cx < F(X1))
    b = b[keep]
    cymin, cxmin = np.maximum(F(0), b[:, 0] - F(Y0)), np.maximum(F(0), b[:, 1] - F(X0))
    cymax, cxmax = np.minimum(F(Y1), b[:, 2]) - F(Y0), np.minimum(F(X1), b[:, 3]) - F(X0)
    cymin, cxmin = np.minimum(cymin, cymax), np.minimum(cxmin, cxmax)
    return (ymin, xmin, ps, ps), np.stack([cymin, cxmin, cymax, cxmax], -1).astype(F)


def draw_geometry(height, width, bboxes, out_shape, anchor_scales, d):
    """The geometric half of preprocess_for_train (:706-722): -> (window (y,x,h,w), boxes scaled to out_shape, flip flag)."""
    bboxes = np.asarray(bboxes, F)
    if d.uniform(0., 1.) < 0.5:                                                       # dan_random_sample (:623-635)
        win, b = dan_random_sample_window(height, width, bboxes, d)
        sy, sx = F(out_shape[0]) / F(win[2]), F(out_shape[1]) / F(win[3])
        b = np.stack([b[:, 0] * F(out_shape[0]) / F(win[2]), b[:, 1] * F(out_shape[1]) / F(win[3]),
                      b[:, 2] * F(out_shape[0]) / F(win[2]), b[:, 3] * F(out_shape[1]) / F(win[3])], -1).astype(F)
    else:                                                                             # data_anchor_sampling (:637-675)
        fh_, fw_ = bboxes[:, 2] - bboxes[:, 0], bboxes[:, 3] - bboxes[:, 1]
        scale = np.maximum(fw_, fh_)
        sel = d.randint(0, len(scale))
        sel_scale = max(scale[sel], F(16.))
        scales = np.asarray(anchor_scales, F)
        anchor_ind = min(int(np.argmax(-np.abs(scales - sel_scale))) + 1, len(scales) - 1) + 1
        target = scales[d.randint(0, anchor_ind)]
        final = d.uniform(target / F(2.), min(F(2.) * np.sqrt(fh_[sel] * fw_[sel]), target * F(2.))) / sel_scale
        patch = min(max(F(out_shape[0]) / final, F(64.)), min(F(height), F(width)) * F(8.))
        win, b = anchor_sample_window(height, width, bboxes, sel, patch, d)
        b = (b * (F(out_shape[0]) / F(win[2]))).astype(F)
    flip = d.uniform(0., 1.) < 0.5                                                     # random_flip_left_right (:609-621)
    if flip:
        b = np.stack([b[:, 0], F(out_shape[1]) - F(1.) - b[:, 3], b[:, 2], F(out_shape[1]) - F(1.) - b[:, 1]], -1).astype(F)
    keep = ((b[:, 2] - b[:, 0]) > F(6.)) & ((b[:, 3] - b[:, 1]) > F(3.))               # :726-732
    return win, b[keep], bool(flip)


# ------------------------------------------------------------------------------------------------ image path
def crop_with_mean_fill(img01, win):
    y, x, h, w = win
    H, W = img01.shape[:2]
    out = np.empty((h, w, 3), F)
    out[...] = np.asarray([m / F(255.) for m in MEANS_RGB], F)
    ys, xs = max(y, 0), max(x, 0)
    ye, xe = min(y + h, H), min(x + w, W)
    if ye > ys and xe > xs:
        out[ys - y:ye - y, xs - x:xe - x] = img01[ys:ye, xs:xe]
    return out


def resize_bilinear_legacy(img, out_h, out_w):
    """tf.image.resize_images(BILINEAR, align_corners=False) of TF 1.8: src = dst * (in / out), no half-pixel offset."""
    H, W = img.shape[:2]
    sy, sx = F(H) / F(out_h), F(W) / F(out_w)
    iy = (np.arange(out_h, dtype=F) * sy).astype(F)
    ix = (np.arange(out_w, dtype=F) * sx).astype(F)
    y0 = np.floor(iy).astype(np.int64); x0 = np.floor(ix).astype(np.int64)
    y1 = np.minimum(y0 + 1, H - 1); x1 = np.minimum(x0 + 1, W - 1)
    ly = (iy - y0.astype(F)).astype(F)[:, None, None]; lx = (ix - x0.astype(F)).astype(F)[None, :, None]
    tl, tr, bl, br = img[y0][:, x0], img[y0][:, x1], img[y1][:, x0], img[y1][:, x1]
    top = (tl + (tr - tl) * lx).astype(F)
    bot = (bl + (br - bl) * lx).astype(F)
    return (top + (bot - top) * ly).astype(F)


def finish(img01, flip):
    """flip, convert_image_dtype(float -> uint8, saturate) = trunc(clip(x * 255.5, 0, 255)), to_float, mean subtraction, BGR (:722-729)."""
    if flip:
        img01 = img01[:, ::-1]
    u8 = np.clip(img01 * F(255.5), F(0), F(255)).astype(np.uint8).astype(F)
    rgb = u8 - np.asarray(MEANS_RGB, F)
    return rgb[..., ::-1].astype(F)


def preprocess_for_train(image_u8, bboxes, out_shape, anchor_scales, d):
    """-> (final image float32 [out_h,out_w,3] BGR mean-subtracted, boxes float32 [k,4] in output pixels, debug dict)."""
    img01 = (image_u8.astype(F) * F(1.0 / 255)).astype(F)                              # convert_image_dtype(uint8 -> float32)
    ops = draw_color_params(d)
    win, boxes, flip = draw_geometry(image_u8.shape[0], image_u8.shape[1], bboxes, out_shape, anchor_scales, d)
    dist, mean = distort_color(img01, ops)
    patch = crop_with_mean_fill(dist, win)
    res = resize_bilinear_legacy(patch, out_shape[0], out_shape[1])
    return finish(res, flip), boxes, {"ops": ops, "win": win, "flip": flip, "contrast_mean": mean}
