"""TF-1.8 built-in semantics restated on PyTorch-CPU (fp32 / fp64), NHWC tensors.

Oracle only (see oracle/__init__.py).  The reference calls these TF built-ins; their
source is not in the reference tree, so the documented TF behaviour is restated and
pinned by hand-computed micro-KATs in tests/test_oracle_tf_semantics.py.
"""
import math

import torch
import torch.nn.functional as F


def same_pad(in_size, k, stride, dilation=1):
    """TF 'SAME' padding (pad_before, pad_after, out) for one spatial dim.

    Follows tf.layers.conv2d(padding='same') as used at net/sfd_net.py:83-88:
    out = ceil(in/stride); total = max((out-1)*stride + k_eff - in, 0); before = total//2.
    """
    k_eff = (k - 1) * dilation + 1
    out = -(-in_size // stride)
    total = max((out - 1) * stride + k_eff - in_size, 0)
    before = total // 2
    return before, total - before, out


def conv2d_same(x, w_hwio, bias=None, stride=1, relu=False):
    """tf.layers.conv2d(padding='same', data_format='channels_last') — net/sfd_net.py:81-89.

    x: [N,H,W,Cin]; w_hwio: [kh,kw,Cin,Cout] (TF kernel layout); cross-correlation.
    Asymmetric SAME padding (e.g. k=3,s=2, even input -> pad (0,1)).
    """
    kh, kw, cin, cout = w_hwio.shape
    n, h, wd, c = x.shape
    assert c == cin
    pt, pb, _ = same_pad(h, kh, stride)
    pl, pr, _ = same_pad(wd, kw, stride)
    xn = x.permute(0, 3, 1, 2)
    xn = F.pad(xn, (pl, pr, pt, pb))
    y = F.conv2d(xn, w_hwio.permute(3, 2, 0, 1), bias, stride=stride)
    if relu:
        y = torch.relu(y)
    return y.permute(0, 2, 3, 1).contiguous()


def conv2d_valid(x, w_hwio, bias=None, stride=1, relu=False):
    """tf.layers.conv2d(padding='valid'): no padding, floor((in - k) / s) + 1 outputs (net/resnet_danet.py:127,161-172)."""
    y = F.conv2d(x.permute(0, 3, 1, 2), w_hwio.permute(3, 2, 0, 1), bias, stride=stride)
    if relu:
        y = torch.relu(y)
    return y.permute(0, 2, 3, 1).contiguous()


def max_pool_3x3_s2_same(x):
    """tf.layers.max_pooling2d(x, [3,3], [2,2], 'same') — net/resnet_danet.py:129: out = ceil(in/2), TF-SAME padding with -inf."""
    n, h, w, c = x.shape
    pt, pb, _ = same_pad(h, 3, 2)
    pl, pr, _ = same_pad(w, 3, 2)
    xn = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb), value=float("-inf"))
    return F.max_pool2d(xn, 3, 2).permute(0, 2, 3, 1).contiguous()


def max_pool_2x2_same(x):
    """tf.layers.max_pooling2d(x, [2,2], [2,2], 'same') — net/sfd_net.py:132-143.

    out = ceil(in/2); odd sizes pad after with -inf.
    """
    n, h, w, c = x.shape
    ph, pw = h % 2, w % 2
    xn = x.permute(0, 3, 1, 2)
    if ph or pw:
        xn = F.pad(xn, (0, pw, 0, ph), value=float("-inf"))
    y = F.max_pool2d(xn, 2, 2)
    return y.permute(0, 2, 3, 1).contiguous()


def avg_pool_2x2_s1_same(x):
    """tf.layers.average_pooling2d(x, (2,2), 1, 'same') — net/danet.py:854.

    out = in; pad (0,1) per dim; divisor counts only VALID taps (TF excludes padding).
    """
    xn = x.permute(0, 3, 1, 2)
    xp = F.pad(xn, (0, 1, 0, 1))
    s = F.avg_pool2d(xp, 2, 1) * 4.0
    ones = F.pad(torch.ones_like(xn[:, :1]), (0, 1, 0, 1))
    cnt = F.avg_pool2d(ones, 2, 1) * 4.0
    return (s / cnt).permute(0, 2, 3, 1).contiguous()


def resize_bilinear_legacy(x, out_h, out_w):
    """tf.image.resize_bilinear(x, size, align_corners=False), TF1 legacy mapping
    (net/pb_net.py:215, net/danet.py:369): src = dst * (in/out); lo = floor(src);
    hi = min(lo+1, in-1); lerp.  Separable; top/bottom blend first then left/right as in TF's
    kernel: top + (bottom - top) * y_lerp.
    """
    n, h, w, c = x.shape

    def axis(inn, out):
        scale = inn / out
        src = torch.arange(out, dtype=torch.float32) * torch.tensor(scale, dtype=torch.float32)
        lo = torch.floor(src).to(torch.int64)
        hi = torch.clamp(lo + 1, max=inn - 1)
        lerp = src - lo.to(torch.float32)
        return lo, hi, lerp.to(x.dtype)

    ylo, yhi, yl = axis(h, out_h)
    xlo, xhi, xl = axis(w, out_w)
    top = x[:, ylo]
    bot = x[:, yhi]
    tl, tr = top[:, :, xlo], top[:, :, xhi]
    bl, br = bot[:, :, xlo], bot[:, :, xhi]
    xl_ = xl.view(1, 1, -1, 1)
    yl_ = yl.view(1, -1, 1, 1)
    t = tl + (tr - tl) * xl_
    b = bl + (br - bl) * xl_
    return t + (b - t) * yl_


def l2_normalize(x, weight):
    """VGG16Backbone.l2_normalize — net/sfd_net.py:68-79:
    x * rsqrt(max(sum_c x^2, 1e-10)) * weight_c."""
    sq = (x * x).sum(-1, keepdim=True)
    inv = torch.rsqrt(torch.clamp(sq, min=1e-10))
    return (x * inv) * weight.view(1, 1, 1, -1)


def maxout_cls(cls_pred, depth, neg_maxout, pos_maxout):
    """Max-out background/face logits — net/sfd_net.py:175-216 (channels_last branch).

    cls_pred [N,H,W,depth*(neg+pos)] -> [N,H,W,depth*2]; channel order inside an anchor group is
    [neg_0..neg_{n-1}, pos_0..pos_{p-1}].  Applied only when pos+neg > 2.
    """
    if pos_maxout + neg_maxout <= 2:
        return cls_pred
    n, h, w, c = cls_pred.shape
    v = cls_pred.reshape(n, h, w, depth, -1)
    pos = v[..., neg_maxout:].amax(-1) if pos_maxout > 1 else v[..., -1]
    neg = v[..., :neg_maxout].amax(-1) if neg_maxout > 1 else v[..., 0]
    return torch.stack([neg, pos], dim=-1).reshape(n, h, w, depth * 2)


def glorot_uniform_(shape, fan_in, fan_out, gen):
    """tf.glorot_uniform_initializer — net/sfd_net.py:65: U(-L, L), L = sqrt(6/(fan_in+fan_out))."""
    lim = math.sqrt(6.0 / (fan_in + fan_out))
    return (torch.rand(shape, generator=gen, dtype=torch.float32) * 2.0 - 1.0) * lim


def batch_norm_train(x, gamma, beta, eps):
    """tf.layers.batch_normalization(training=True, fused) over the channel axis of NHWC —
    net/sfd_net.py:91-119. Returns (y, batch_mean, batch_var[biased]).  (The moving_variance update of TF1's fused path uses the
    Bessel-corrected variance: batch_norm_moving_variance below.)"""
    mean = x.mean((0, 1, 2))
    var = x.var((0, 1, 2), unbiased=False)
    y = (x - mean) * torch.rsqrt(var + eps) * gamma + beta
    return y, mean, var


def batch_norm_moving_variance(moving_var, batch_var_biased, count, momentum):
    """Moving-variance update of TF1's fused batch norm: the batch variance fed to the moving average is var * M / (M - 1)."""
    unbiased = batch_var_biased * (count / (count - 1.0)) if count > 1 else batch_var_biased
    return moving_var * momentum + unbiased * (1.0 - momentum)


def batch_norm_infer(x, gamma, beta, mean, var, eps):
    return (x - mean) * torch.rsqrt(var + eps) * gamma + beta


# ---------------------------------------------------------------------------------------------------------------
# bf16 storage emulation (what the MI355X build stores between kernels): used by the parity tests so that the only
# remaining differences to the HIP path are accumulation order and ties at the ReLU boundary.
# storage type emulated by round_bf16: bf16 (default build) or torch.float16 (the fp16 build, tests set it explicitly)
EMULATE_DTYPE = torch.bfloat16


class _RoundBF16(torch.autograd.Function):
    """forward: round to bf16 (kept in fp32 container) if fwd; backward: round the incoming gradient if bwd."""

    @staticmethod
    def forward(ctx, x, fwd, bwd):
        ctx.bwd = bwd
        return x.to(EMULATE_DTYPE).to(x.dtype) if fwd else x.clone()

    @staticmethod
    def backward(ctx, g):
        return (g.to(EMULATE_DTYPE).to(g.dtype) if ctx.bwd else g), None, None


def round_bf16(x, fwd=True, bwd=True):
    return _RoundBF16.apply(x, fwd, bwd)
