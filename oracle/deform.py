"""Deformable convolution oracle (NCHW, fp32/fp64) — restates cpp/Deform/deform_conv.cu and the
orchestration in cpp/Deform/deform_conv.cc.  Oracle only (see oracle/__init__.py).

Forward  : deformable_im2col_gpu_kernel (deform_conv.cu:229-275) +
           deformable_im2col_bilinear (deform_conv.cu:91-126) + GEMM (deform_conv.cc:489-517).
Backward : deformable_col2im_coord_gpu_kernel (:335-389) + get_coordinate_weight (:177-221),
           deformable_col2im_gpu_kernel (:281-328) + get_gradient_weight (:130-173),
           dW = dY . col^T (deform_conv.cc:760-767).

Parity status: the reference's own test (cpp/Deform/test_deform_conv.py) only prints; pinned here by
the identity KAT "zero offsets == plain SAME conv"  and by finite differences.
"""
import torch

from .tf_ops import same_pad


def _window(in_size, k, stride, padding):
    """GetWindowedOutputSize as DeformConvOp calls it (deform_conv.cc:473-479: the UNDILATED kernel size): (pad_before, out)."""
    if padding == "SAME":
        before, _, out = same_pad(in_size, k, stride)
        return before, out
    assert padding == "VALID", padding
    return 0, -(-(in_size - k + 1) // stride)


def _sample_setup(x, offset, kh, kw, stride, dilation, dg, padding="SAME"):
    """Common sampling geometry. x [B,C,H,W], offset [B, dg*2*kh*kw, Ho, Wo].

    Returns dict with per-(b,g,tap,ho,wo) tensors mirroring the kernel's arithmetic:
      h_im/w_im   : absolute sample coordinate  h_in + i*dil + off   (deform_conv.cu:261-262)
      map_h/map_w : coordinate relative to (h_in,w_in): i*dil + off   (:264-265)
    pad = TF SAME pad_before from the UNDILATED kernel (deform_conv.cc:473-479).
    """
    B, C, H, W = x.shape
    pad_h, Ho = _window(H, kh, stride, padding)
    pad_w, Wo = _window(W, kw, stride, padding)
    assert offset.shape == (B, dg * 2 * kh * kw, Ho, Wo), (offset.shape, (B, dg * 2 * kh * kw, Ho, Wo))
    off = offset.reshape(B, dg, kh * kw, 2, Ho, Wo)
    off_h, off_w = off[:, :, :, 0], off[:, :, :, 1]            # [B,dg,T,Ho,Wo]
    dt = x.dtype
    h_in = (torch.arange(Ho) * stride - pad_h).view(1, 1, 1, Ho, 1)
    w_in = (torch.arange(Wo) * stride - pad_w).view(1, 1, 1, 1, Wo)
    ti = (torch.arange(kh * kw) // kw * dilation).view(1, 1, -1, 1, 1)
    tj = (torch.arange(kh * kw) % kw * dilation).view(1, 1, -1, 1, 1)
    # (h_in + i*dil) is integer arithmetic in the kernel, then + offset in DType
    h_im = (h_in + ti).to(dt) + off_h
    w_im = (w_in + tj).to(dt) + off_w
    map_h = ti.to(dt) + off_h
    map_w = tj.to(dt) + off_w
    return dict(B=B, C=C, H=H, W=W, Ho=Ho, Wo=Wo, h_in=h_in, w_in=w_in, h_im=h_im, w_im=w_im,
                map_h=map_h, map_w=map_w)


def deform_im2col(x, offset, kh, kw, stride=1, dilation=1, dg=1, padding="SAME"):
    """Returns col [B, C, kh*kw, Ho, Wo] (differentiable w.r.t. x through torch autograd)."""
    g = _sample_setup(x, offset, kh, kw, stride, dilation, dg, padding)
    B, C, H, W, Ho, Wo = g["B"], g["C"], g["H"], g["W"], g["Ho"], g["Wo"]
    cpg = C // dg
    inb = (g["h_im"] >= 0) & (g["w_im"] >= 0) & (g["h_im"] < H) & (g["w_im"] < W)
    # bilinear relative to the (h_in, w_in) window; cur_height = H - h_in  (deform_conv.cu:266-268)
    h_low = torch.floor(g["map_h"])
    w_low = torch.floor(g["map_w"])
    cur_h = (H - g["h_in"]).to(x.dtype)
    cur_w = (W - g["w_in"]).to(x.dtype)
    clamp_h = h_low >= cur_h - 1
    clamp_w = w_low >= cur_w - 1
    h_low = torch.where(clamp_h, cur_h - 1 + torch.zeros_like(h_low), h_low)
    w_low = torch.where(clamp_w, cur_w - 1 + torch.zeros_like(w_low), w_low)
    mh = torch.where(clamp_h, h_low, g["map_h"])
    mw = torch.where(clamp_w, w_low, g["map_w"])
    h_high = torch.where(clamp_h, h_low, h_low + 1)
    w_high = torch.where(clamp_w, w_low, w_low + 1)
    lh, lw = mh - h_low, mw - w_low
    hh, hw = 1 - lh, 1 - lw
    # absolute integer rows/cols (valid only where inb)
    ah_lo = (h_low + g["h_in"]).long().clamp(0, H - 1)
    ah_hi = (h_high + g["h_in"]).long().clamp(0, H - 1)
    aw_lo = (w_low + g["w_in"]).long().clamp(0, W - 1)
    aw_hi = (w_high + g["w_in"]).long().clamp(0, W - 1)
    xg = x.reshape(B, dg, cpg, H * W)

    def gather(ah, aw):
        idx = (ah * W + aw).reshape(B, dg, 1, -1).expand(B, dg, cpg, -1)
        return torch.gather(xg, 3, idx).reshape(B, dg, cpg, kh * kw, Ho, Wo)

    v1, v2, v3, v4 = gather(ah_lo, aw_lo), gather(ah_lo, aw_hi), gather(ah_hi, aw_lo), gather(ah_hi, aw_hi)
    w1, w2, w3, w4 = (hh * hw).unsqueeze(2), (hh * lw).unsqueeze(2), (lh * hw).unsqueeze(2), (lh * lw).unsqueeze(2)
    val = w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4
    val = torch.where(inb.unsqueeze(2), val, torch.zeros_like(val))
    return val.reshape(B, C, kh * kw, Ho, Wo)


def deform_conv_forward(x, w_oihw, offset, stride=1, dilation=1, dg=1, padding="SAME", num_groups=1):
    """DeformConvOp (deform_conv.cc:392-535): out[b][g] = W[g][Co/G, Cin/G*kh*kw] . col[b][g] — the batched matmul over `group_`
    contiguous channel groups of :487-515 (weight viewed [group, M, K], column buffer [group, K, N]).
    x [B,Cin,H,W], w [Cout,Cin/G,kh,kw], offset [B,2*kh*kw*dg,Ho,Wo] -> [B,Cout,Ho,Wo]."""
    co, cig, kh, kw = w_oihw.shape
    col = deform_im2col(x, offset, kh, kw, stride, dilation, dg, padding)
    B, C, T, Ho, Wo = col.shape
    G = num_groups
    assert C == cig * G and co % G == 0
    colg = col.reshape(B, G, cig * T, Ho * Wo)
    wg = w_oihw.reshape(G, co // G, cig * T)
    return torch.einsum("gok,bgkn->bgon", wg, colg).reshape(B, co, Ho, Wo)


def _coord_weight_parts(g, H, W):
    """get_coordinate_weight geometry (deform_conv.cu:177-221) on absolute coords inv_h/inv_w.
    Out-of-range samples are mapped to -1 and yield weight 0 (:371-373, :181-185)."""
    inv_h, inv_w = g["h_im"], g["w_im"]
    oob = (inv_h < 0) | (inv_w < 0) | (inv_h >= H) | (inv_w >= W)
    ih = torch.where(oob, torch.zeros_like(inv_h), inv_h)
    iw = torch.where(oob, torch.zeros_like(inv_w), inv_w)
    h_low = torch.trunc(ih)
    w_low = torch.trunc(iw)
    ch, cw = h_low >= H - 1, w_low >= W - 1
    h_low = torch.where(ch, torch.full_like(h_low, H - 1), h_low)
    w_low = torch.where(cw, torch.full_like(w_low, W - 1), w_low)
    ih = torch.where(ch, h_low, ih)
    iw = torch.where(cw, w_low, iw)
    h_high = torch.where(ch, h_low, h_low + 1)
    w_high = torch.where(cw, w_low, w_low + 1)
    return oob, ih, iw, h_low, w_low, h_high, w_high


def deform_conv_backward(x, w_oihw, offset, dy, stride=1, dilation=1, dg=1, padding="SAME"):
    """DeformConvBackpropOp (deform_conv.cc:635-771): returns (dx, dw, doffset), explicit formulas (num_groups = 1)."""
    co, ci, kh, kw = w_oihw.shape
    T = kh * kw
    g = _sample_setup(x, offset, kh, kw, stride, dilation, dg, padding)
    B, C, H, W, Ho, Wo = g["B"], g["C"], g["H"], g["W"], g["Ho"], g["Wo"]
    cpg = C // dg
    # col_grad = W^T . dY   [B, C, T, Ho, Wo]   (deform_conv.cc:736)
    colg = torch.einsum("ok,bon->bkn", w_oihw.reshape(co, ci * T), dy.reshape(B, co, Ho * Wo)).reshape(B, dg, cpg, T, Ho, Wo)
    # ---- dOffset (deformable_col2im_coord_gpu_kernel)
    oob, ih, iw, h_low, w_low, h_high, w_high = _coord_weight_parts(g, H, W)
    xg = x.reshape(B, dg, cpg, H * W)

    def gather(ah, aw):
        idx = (ah.long().clamp(0, H - 1) * W + aw.long().clamp(0, W - 1)).reshape(B, dg, 1, -1).expand(B, dg, cpg, -1)
        return torch.gather(xg, 3, idx).reshape(B, dg, cpg, T, Ho, Wo)

    v_ll, v_lh, v_hl, v_hh = gather(h_low, w_low), gather(h_low, w_high), gather(h_high, w_low), gather(h_high, w_high)
    a_w = (w_low + 1 - iw).unsqueeze(2)
    b_w = (iw - w_low).unsqueeze(2)
    a_h = (h_low + 1 - ih).unsqueeze(2)
    b_h = (ih - h_low).unsqueeze(2)
    wgt_h = -1 * a_w * v_ll + -1 * b_w * v_lh + a_w * v_hl + b_w * v_hh       # bp_dir == 0
    wgt_w = -1 * a_h * v_ll + a_h * v_lh + -1 * b_h * v_hl + b_h * v_hh       # bp_dir == 1
    zero = torch.zeros_like(wgt_h)
    wgt_h = torch.where(oob.unsqueeze(2), zero, wgt_h)
    wgt_w = torch.where(oob.unsqueeze(2), zero, wgt_w)
    d_off_h = (wgt_h * colg).sum(2)                                            # [B,dg,T,Ho,Wo]
    d_off_w = (wgt_w * colg).sum(2)
    doffset = torch.stack([d_off_h, d_off_w], dim=3).reshape(B, dg * 2 * T, Ho, Wo)
    # ---- dX (deformable_col2im_gpu_kernel): scatter col_grad * get_gradient_weight to the <=4 corners.
    inv_h, inv_w = g["h_im"], g["w_im"]
    empty = (inv_h < 0) | (inv_h > H) | (inv_w < 0) | (inv_w > W)
    ah = torch.clamp(inv_h, min=0)
    aw = torch.clamp(inv_w, min=0)
    hl = torch.trunc(ah)
    wl = torch.trunc(aw)
    ch, cw = hl >= H - 1, wl >= W - 1
    hl = torch.where(ch, torch.full_like(hl, H - 1), hl)
    wl = torch.where(cw, torch.full_like(wl, W - 1), wl)
    ah = torch.where(ch, hl, ah)
    aw = torch.where(cw, wl, aw)
    hh_ = torch.where(ch, hl, hl + 1)
    wh_ = torch.where(cw, wl, wl + 1)
    dx = torch.zeros(B, dg, cpg, H * W, dtype=x.dtype)
    # corner weights exactly as get_gradient_weight's four cases; when low==high (clamped) only the first
    # matching branch (h == low, w == low) fires in the reference's if/else-if chain.
    corners = [
        (hl, wl, (hl + 1 - ah) * (wl + 1 - aw), None),
        (hl, wh_, (hl + 1 - ah) * (aw + 1 - wh_), cw),
        (hh_, wl, (ah + 1 - hh_) * (wl + 1 - aw), ch),
        (hh_, wh_, (ah + 1 - hh_) * (aw + 1 - wh_), ch | cw),
    ]
    # the reference visits integer neighbours (cur_h+dy, cur_w+dx) with |inv - n| < 1; this keeps exactly the
    # corners whose coordinate differs from the sample by < 1 (deform_conv.cu:314-326).
    for (rh, rw, wgt, dup) in corners:
        ok = (~empty) & ((inv_h - rh).abs() < 1) & ((inv_w - rw).abs() < 1) & (rh >= 0) & (rh < H) & (rw >= 0) & (rw < W)
        if dup is not None:
            ok = ok & (~dup)
        contrib = torch.where(ok.unsqueeze(2), wgt.unsqueeze(2) * colg, torch.zeros_like(colg))
        idx = (rh.long().clamp(0, H - 1) * W + rw.long().clamp(0, W - 1)).reshape(B, dg, 1, -1).expand(B, dg, cpg, -1)
        dx.scatter_add_(3, idx, contrib.reshape(B, dg, cpg, -1))
    dx = dx.reshape(B, C, H, W)
    # ---- dW = sum_b dY_b . col_b^T  (deform_conv.cc:757-768)
    col = deform_im2col(x, offset, kh, kw, stride, dilation, dg, padding).reshape(B, C * T, Ho * Wo)
    dw = torch.einsum("bon,bkn->ok", dy.reshape(B, co, Ho * Wo), col).reshape(co, ci, kh, kw)
    return dx, dw, doffset


# ------------------------------------------------------------------------------------------------------------------
# Deformable PS-ROI pooling — cpp/Deform/deform_psroi_pooling_op_gpu.cu:47-125 (forward), :187-300 (backward); numpy, one output
# element at a time, the float / double promotions of the CUDA source kept (float data, double-typed literals, C round()).
def _psroi_bin(rois, trans, n, ctop, ph, pw, at, backward):
    import numpy as np
    f = np.float32
    r = rois[n]
    rnd = lambda v: f(np.floor(abs(float(v)) + 0.5) * (1.0 if v >= 0 else -1.0))             # C round(): half away from zero
    scale = f(at["spatial_scale"])
    roi_start_w = f(np.float64(f(rnd(r[1]) * scale)) - 0.5)
    roi_start_h = f(np.float64(f(rnd(r[2]) * scale)) - 0.5)
    roi_end_w = f(np.float64(f(f(np.float64(rnd(r[3])) + 1.) * scale)) - 0.5)
    roi_end_h = f(np.float64(f(f(np.float64(rnd(r[4])) + 1.) * scale)) - 0.5)
    if backward:
        roi_w, roi_h = max(f(roi_end_w - roi_start_w), f(0.1)), max(f(roi_end_h - roi_start_h), f(0.1))
    else:
        roi_w, roi_h = f(max(np.float64(f(roi_end_w - roi_start_w)), 0.1)), f(max(np.float64(f(roi_end_h - roi_start_h)), 0.1))
    P, spp, part, G = at["pooled_size"], at["sample_per_part"], at["part_size"], at["group_size"]
    bin_h, bin_w = f(roi_h / f(P)), f(roi_w / f(P))
    sub_h, sub_w = f(bin_h / f(spp)), f(bin_w / f(spp))
    part_h = int(np.floor(f(f(f(ph) / f(P)) * f(part))))
    part_w = int(np.floor(f(f(f(pw) / f(P)) * f(part))))
    ncls = at["num_classes"]
    class_id = ctop // (at["output_dim"] // ncls)
    if at["no_trans"]:
        tx = ty = f(0)
    else:
        tx = f(trans[n, class_id * 2, part_h, part_w] * f(at["trans_std"]))
        ty = f(trans[n, class_id * 2 + 1, part_h, part_w] * f(at["trans_std"]))
    wstart = f(f(f(pw) * bin_w) + roi_start_w)
    wstart = f(wstart + f(tx * roi_w))
    hstart = f(f(f(ph) * bin_h) + roi_start_h)
    hstart = f(hstart + f(ty * roi_h))
    gw = min(max(int(np.floor(f(f(f(pw) * f(G)) / f(P)))), 0), G - 1)
    gh = min(max(int(np.floor(f(f(f(ph) * f(G)) / f(P)))), 0), G - 1)
    return dict(b=int(r[0]), roi_w=roi_w, roi_h=roi_h, wstart=wstart, hstart=hstart, sub_w=sub_w, sub_h=sub_h, part_h=part_h, part_w=part_w,
                class_id=class_id, gw=gw, gh=gh)


def _psroi_samples(bn, at, H, W):
    import numpy as np
    f = np.float32
    for ih in range(at["sample_per_part"]):
        for iw in range(at["sample_per_part"]):
            w = f(bn["wstart"] + f(f(iw) * bn["sub_w"]))
            h = f(bn["hstart"] + f(f(ih) * bn["sub_h"]))
            if float(w) < -0.5 or float(w) > W - 0.5 or float(h) < -0.5 or float(h) > H - 0.5:
                continue
            w = f(min(max(float(w), 0.), W - 1.))
            h = f(min(max(float(h), 0.), H - 1.))
            yield w, h


def deform_psroi_pool_forward(data, rois, trans, **at):
    """data [B,C,H,W], rois [R,5], trans [R,2*ncls,part,part] (float32 numpy) -> (top_data, mapping_channel) [R,output_dim,P,P]."""
    import numpy as np
    f = np.float32
    B, C, H, W = data.shape
    R, P, OD, G = rois.shape[0], at["pooled_size"], at["output_dim"], at["group_size"]
    at = dict(at, num_classes=1 if at["no_trans"] else trans.shape[1] // 2)
    top, cnt = np.zeros((R, OD, P, P), f), np.zeros((R, OD, P, P), f)
    for n in range(R):
        for ctop in range(OD):
            for ph in range(P):
                for pw in range(P):
                    bn = _psroi_bin(rois, trans, n, ctop, ph, pw, at, False)
                    d = data[bn["b"], (ctop * G + bn["gh"]) * G + bn["gw"]]
                    s, c = f(0), 0
                    for w, h in _psroi_samples(bn, at, H, W):
                        x1, x2, y1, y2 = int(np.floor(w)), int(np.ceil(w)), int(np.floor(h)), int(np.ceil(h))
                        dx, dy = f(w - f(x1)), f(h - f(y1))
                        o = f(1)
                        val = f(f(f(f(o - dx) * f(o - dy)) * d[y1, x1]) + f(f(f(o - dx) * dy) * d[y2, x1]))
                        val = f(val + f(f(dx * f(o - dy)) * d[y1, x2]))
                        val = f(val + f(f(dx * dy) * d[y2, x2]))
                        s = f(s + val)
                        c += 1
                    top[n, ctop, ph, pw] = f(0) if c == 0 else f(s / f(c))
                    cnt[n, ctop, ph, pw] = c
    return top, cnt


def deform_psroi_pool_backward(data, rois, trans, top_count, top_diff, **at):
    """-> (data_diff [B,C,H,W], trans_diff like trans); float64 accumulation (the device uses fp32 atomics in arbitrary order)."""
    import numpy as np
    f = np.float32
    B, C, H, W = data.shape
    R, P, OD, G = rois.shape[0], at["pooled_size"], at["output_dim"], at["group_size"]
    at = dict(at, num_classes=1 if at["no_trans"] else trans.shape[1] // 2)
    dd = np.zeros(data.shape, np.float64)
    dt = np.zeros(trans.shape, np.float64)
    for n in range(R):
        for ctop in range(OD):
            for ph in range(P):
                for pw in range(P):
                    if top_count[n, ctop, ph, pw] <= 0:
                        continue
                    bn = _psroi_bin(rois, trans, n, ctop, ph, pw, at, True)
                    dv = f(top_diff[n, ctop, ph, pw] / top_count[n, ctop, ph, pw])
                    c = (ctop * G + bn["gh"]) * G + bn["gw"]
                    d = data[bn["b"], c]
                    for w, h in _psroi_samples(bn, at, H, W):
                        x0, x1, y0, y1 = int(np.floor(w)), int(np.ceil(w)), int(np.floor(h)), int(np.ceil(h))
                        dx, dy = f(w - f(x0)), f(h - f(y0))
                        o = f(1)
                        dd[bn["b"], c, y0, x0] += f(f(f(o - dx) * f(o - dy)) * dv)
                        dd[bn["b"], c, y1, x0] += f(f(f(o - dx) * dy) * dv)
                        dd[bn["b"], c, y0, x1] += f(f(dx * f(o - dy)) * dv)
                        dd[bn["b"], c, y1, x1] += f(f(dx * dy) * dv)
                        if at["no_trans"]:
                            continue
                        U00, U01, U10, U11 = d[y0, x0], d[y1, x0], d[y0, x1], d[y1, x1]
                        gx = f(f(f(f(f(f(U11 * dy) + f(U10 * f(o - dy))) - f(U01 * dy)) - f(U00 * f(o - dy))) * f(at["trans_std"])) * dv)
                        gx = f(gx * bn["roi_w"])
                        gy = f(f(f(f(f(f(U11 * dx) + f(U01 * f(o - dx))) - f(U10 * dx)) - f(U00 * f(o - dx))) * f(at["trans_std"])) * dv)
                        gy = f(gy * bn["roi_h"])
                        dt[n, bn["class_id"] * 2, bn["part_h"], bn["part_w"]] += gx
                        dt[n, bn["class_id"] * 2 + 1, bn["part_h"], bn["part_w"]] += gy
    return dd.astype(f), dt.astype(f)
