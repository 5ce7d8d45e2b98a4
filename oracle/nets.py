"""The four detector graphs restated on PyTorch-CPU, NHWC (channels_last) — oracle only.

Follows net/sfd_net.py, net/pb_net.py, net/danet.py, net/danet_deform.py and
utility/custom_op.py:128-146 of the reference.  Variables live in a flat dict keyed by the
TF variable names (without the model scope), e.g. "conv1/conv1_1/conv2d/kernel" [kh,kw,Cin,Cout].

`Params(create=True, seed=...)` creates variables on first use with the reference's initialisers
(glorot-uniform kernels net/sfd_net.py:65, zero biases, l2-norm scale 10/8/5), which is how the
synthetic weights of SURVEY §8(d) are made.
"""
import torch
import torch.nn.functional as F

from . import tf_ops as T
from .deform import deform_conv_forward, deform_im2col


class Params:
    def __init__(self, tensors=None, create=False, seed=20180817, dtype=torch.float32, emulate_bf16=False):
        # emulate_bf16: round conv weights (straight-through), stored activations and their gradients to bf16, as the
        # MI355X build stores them (fp32 accumulate everywhere) — used only to tighten parity tolerances.
        self.emulate_bf16 = emulate_bf16
        # impose (tests/test_grad_parity_gpu.py): {"relu": {kernel name: activation of the run under test}, "pools": [pre-pool activations in
        # call order]} — the DISCRETE decisions (ReLU sign, 2x2 arg-max) are taken from those tensors instead of from this graph's own
        # values, so that a comparison of gradients is not dominated by pre-activations within rounding of zero landing on different sides
        self.impose = None
        self._pool_no = 0
        self.t = dict(tensors or {})
        self.create = create
        self.gen = torch.Generator().manual_seed(seed)
        self.dtype = dtype

    def get(self, name, shape, init):
        if name not in self.t:
            if not self.create:
                raise KeyError(name)
            if init == "glorot":                       # HWIO conv kernel
                kh, kw, ci, co = shape
                v = T.glorot_uniform_(shape, kh * kw * ci, kh * kw * co, self.gen)
            elif init == "glorot_oihw":                # deform kernel (Cout,Cin,kh,kw): TF computes fans from
                co, ci, kh, kw = shape                  # the shape as given: receptive=co*ci, in=kh, out=kw
                v = T.glorot_uniform_(shape, co * ci * kh, co * ci * kw, self.gen)
            elif init == "zeros":
                v = torch.zeros(shape)
            elif isinstance(init, (int, float)):
                v = torch.full(shape, float(init))
            else:
                raise ValueError(init)
            self.t[name] = v.to(self.dtype)
        v = self.t[name]
        assert tuple(v.shape) == tuple(shape), (name, tuple(v.shape), tuple(shape))
        return v


def _relu(P, y, scope):
    """ReLU of a layer's pre-activation; with P.impose the sign pattern comes from the recorded activation of that layer."""
    if P.impose is not None and scope + "/kernel" in P.impose["relu"]:
        ref = P.impose["relu"][scope + "/kernel"]
        assert ref.shape == y.shape, (scope, tuple(ref.shape), tuple(y.shape))
        return y * (ref > 0).to(y.dtype)
    return torch.relu(y)


def _maxout(P, c, depth, neg_maxout, pos_maxout, scope):
    """T.maxout_cls; with P.impose the arg-max among the background logits comes from the recorded head output (depth 1)."""
    ref = None if P.impose is None else P.impose.get("maxout", {}).get(scope + "/kernel")
    if ref is None or pos_maxout + neg_maxout <= 2:
        return T.maxout_cls(c, depth, neg_maxout, pos_maxout)
    assert depth == 1 and ref.shape == c.shape, (scope, tuple(ref.shape), tuple(c.shape))
    pick = lambda lo, hi: torch.gather(c[..., lo:hi], -1, ref[..., lo:hi].argmax(-1, keepdim=True))
    neg = pick(0, neg_maxout) if neg_maxout > 1 else c[..., :1]
    pos = pick(neg_maxout, neg_maxout + pos_maxout) if pos_maxout > 1 else c[..., -1:]
    return torch.cat([neg, pos], dim=-1)


def max_pool(P, x):
    """tf.layers.max_pooling2d([2,2],[2,2],'same') (net/sfd_net.py:132-143); with P.impose the arg-max positions come from the recorded
    pre-pool activation (same window scan order: the first maximum wins)."""
    if P.impose is None:
        return T.max_pool_2x2_same(x)
    ref = P.impose["pools"][P._pool_no]
    P._pool_no += 1
    assert ref.shape == x.shape, (tuple(ref.shape), tuple(x.shape))
    n, h, w, c = x.shape
    ph, pw = h % 2, w % 2
    rn = F.pad(ref.permute(0, 3, 1, 2), (0, pw, 0, ph), value=float("-inf"))
    xn = F.pad(x.permute(0, 3, 1, 2), (0, pw, 0, ph))
    _, idx = F.max_pool2d(rn, 2, 2, return_indices=True)
    y = torch.gather(xn.flatten(2), 2, idx.flatten(2)).view(idx.shape)
    return y.permute(0, 2, 3, 1).contiguous()


def conv(P, x, filters, ksize, stride, scope, relu, init="glorot"):
    """tf.layers.conv2d(name=<scope>) with bias; kernel var '<scope>/kernel', bias '<scope>/bias'."""
    kh, kw = ksize
    w = P.get(scope + "/kernel", (kh, kw, x.shape[-1], filters), init)
    b = P.get(scope + "/bias", (filters,), "zeros")
    if P.emulate_bf16:
        w = T.round_bf16(w, True, False)                      # bf16 packed weights, fp32 master gradient
        head = scope.rsplit("/", 1)[-1].startswith(("loc_", "cls_"))
        y = T.conv2d_same(x, w, b, stride=stride, relu=False)
        if head:                                              # head convs emit fp32; their incoming gradient is cast to bf16
            return T.round_bf16(y, False, True)
        y = T.round_bf16(y, False, True) if relu else y       # gradient w.r.t. the pre-activation is stored in bf16
        y = _relu(P, y, scope) if relu else y
        return T.round_bf16(y, True, not relu)                # stored activation is bf16
    if relu and P.impose is not None:
        return _relu(P, T.conv2d_same(x, w, b, stride=stride, relu=False), scope)
    return T.conv2d_same(x, w, b, stride=stride, relu=relu)


def conv_relu(P, x, filters, ksize, stride, scope):
    """VGG16Backbone.conv_relu — net/sfd_net.py:81-89 (variables under '<scope>/conv2d/')."""
    return conv(P, x, filters, ksize, stride, scope + "/conv2d", relu=True)


def conv_block(P, x, n, filters, name):
    """net/sfd_net.py:121-125."""
    for i in range(1, n + 1):
        x = conv_relu(P, x, filters, (3, 3), 1, "{0}/{0}_{1}".format(name, i))
    return x


def get_featmaps(P, x):
    """VGG16Backbone.get_featmaps — net/sfd_net.py:127-156 (identical in pb_net/danet)."""
    feats = []
    x = conv_block(P, x, 2, 64, "conv1")
    x = max_pool(P, x)
    x = conv_block(P, x, 2, 128, "conv2")
    x = max_pool(P, x)
    x = conv_block(P, x, 3, 256, "conv3")
    l2n = (lambda a, g: T.round_bf16(T.l2_normalize(a, g), True, True)) if P.emulate_bf16 else T.l2_normalize
    feats.append(l2n(x, P.get("l2_norm_layer_3/weight", (256,), 10.0)))
    x = max_pool(P, x)
    x = conv_block(P, x, 3, 512, "conv4")
    feats.append(l2n(x, P.get("l2_norm_layer_4/weight", (512,), 8.0)))
    x = max_pool(P, x)
    x = conv_block(P, x, 3, 512, "conv5")
    feats.append(l2n(x, P.get("l2_norm_layer_5/weight", (512,), 5.0)))
    x = max_pool(P, x)
    x = conv_relu(P, x, 1024, (3, 3), 1, "fc6")
    x = conv_relu(P, x, 1024, (1, 1), 1, "fc7")
    feats.append(x)
    x = conv_relu(P, x, 256, (1, 1), 1, "additional_layers/conv6_1")
    x = conv_relu(P, x, 512, (3, 3), 2, "additional_layers/conv6_2")
    feats.append(x)
    x = conv_relu(P, x, 128, (1, 1), 1, "additional_layers/conv7_1")
    x = conv_relu(P, x, 256, (3, 3), 2, "additional_layers/conv7_2")
    feats.append(x)
    return feats


def predict_module(P, feats, pos_maxout, neg_maxout, depth, name, shared_conv=False):
    """multibox_head (net/sfd_net.py:159-219, name='multibox_head'), PB get_predict_module
    (net/pb_net.py:230-290) and DAN get_predict_module with shared conv (net/danet.py:469-532)."""
    locs, clss = [], []
    for i, f in enumerate(feats):
        if shared_conv:
            f = conv_relu(P, f, f.shape[-1], (3, 3), 1, "{}/shared_conv_{}".format(name, i))
        locs.append(conv(P, f, depth[i] * 4, (3, 3), 1, "{}/loc_{}".format(name, i), relu=False))
        c = conv(P, f, depth[i] * (pos_maxout[i] + neg_maxout[i]), (3, 3), 1, "{}/cls_{}".format(name, i), relu=False)
        clss.append(_maxout(P, c, depth[i], neg_maxout[i], pos_maxout[i], "{}/cls_{}".format(name, i)))
    return locs, clss


def build_lfpn(P, feats, skip_last=3, name="lfpn", fused_channels=None):
    """PB build_lfpn (net/pb_net.py:185-226; fused_conv -> down_channels) and DAN build_lfpn
    (net/danet.py:339-380; fused_conv -> 256). No activation on any conv; the running
    `up_sampling` is the SUM (lateral + upsampled), not the fused output."""
    outs = []
    up = None
    for ind in range(skip_last, 0, -1):
        sc = "{}/fpn_{}".format(name, ind - 1)
        down = feats[ind - 1].shape[-1]
        if up is None:
            up = feats[ind]
        up = conv(P, up, down, (1, 1), 1, sc + "/upsample_conv", relu=False)
        lat = conv(P, feats[ind - 1], down, (1, 1), 1, sc + "/lateral", relu=False)
        up = T.resize_bilinear_legacy(up, lat.shape[1], lat.shape[2])
        up = lat + up
        if P.emulate_bf16:
            up = T.round_bf16(up, True, True)                 # the merged map is a stored bf16 activation on the MI355X build
        outs.append(conv(P, up, fused_channels or down, (3, 3), 1, sc + "/fused_conv", relu=False))
    return list(reversed(outs)) + list(feats[skip_last:])


def build_bi_lfpn(P, feats, skip_last=3, name="lfpn"):
    """net/danet.py:191-249: every level but the last = fused_conv3x3->256 ( feat + resize(lateral1x1(next coarser level)) )."""
    outs = []
    for ind, f in enumerate(feats[:-1]):
        sc = "{}/fpn_{}".format(name, ind)
        up = conv(P, feats[ind + 1], f.shape[-1], (1, 1), 1, sc + "/lateral", relu=False)
        m = f + T.resize_bilinear_legacy(up, f.shape[1], f.shape[2])
        if P.emulate_bf16:
            m = T.round_bf16(m, True, True)
        outs.append(conv(P, m, 256, (3, 3), 1, sc + "/fused_conv", relu=False))
    return outs + [feats[-1]]


def build_reverse_lfpn(P, feats, skip_last=3, name="reverse_lfpn"):
    """net/danet.py:382-412: down = conv3x3/s2(down or feats[ind]) -> C_{ind+1}; lat = conv1x1(feats[ind+1]); down = lat + down;
    out_ind = conv3x3(down) -> 256.  Returns [feats[0]] + outs + feats[skip_last+1:]."""
    outs, down = [], None
    for ind in range(0, skip_last):
        sc = "{}/reverse_fpn_{}".format(name, ind)
        dc = feats[ind + 1].shape[-1]
        if down is None:
            down = feats[ind]
        down = conv(P, down, dc, (3, 3), 2, sc + "/downsample_conv", relu=False)
        lat = conv(P, feats[ind + 1], dc, (1, 1), 1, sc + "/lateral", relu=False)
        down = lat + down
        if P.emulate_bf16:
            down = T.round_bf16(down, True, True)
        outs.append(conv(P, down, 256, (3, 3), 1, sc + "/fused_conv", relu=False))
    return [feats[0]] + outs + list(feats[skip_last + 1:])


def context_pred_module(P, feats):
    """PyramidBox CPM — net/pb_net.py:158-183 (hard-coded 1024 channels)."""
    def block(x, nc, last_div, name):
        x = conv_relu(P, x, nc, (3, 3), 1, name + "/conv1")
        x = conv_relu(P, x, nc // 4, (3, 3), 1, name + "/conv2")
        return conv_relu(P, x, nc // last_div, (3, 3), 1, name + "/conv3")
    outs = []
    for i, f in enumerate(feats):
        nc = 1024
        b1 = block(f, nc, 4, "cpm/branch{}_1".format(i))
        b2 = block(f, nc, 4, "cpm/branch{}_2".format(i))
        b2_1 = block(b2, nc, 8, "cpm/branch{}_2_1".format(i))
        b2_2_1 = block(b2, nc, 8, "cpm/branch{}_2_2_1".format(i))
        b2_2_2 = block(b2_2_1, nc, 8, "cpm/branch{}_2_2_2".format(i))
        outs.append(torch.cat([b1, b2_1, b2_2_2], dim=-1))
    return outs


def se_inception_block_v1(P, x, name):
    """DAN context module V1 — net/danet.py:842-918."""
    c = x.shape[-1]
    cr = lambda inp, f, k, n: conv(P, inp, f, k, 1, name + "/" + n, relu=True)
    b1 = cr(x, 64, (1, 1), "branch1_conv_1x1")
    ap = T.avg_pool_2x2_s1_same(x)
    if P.emulate_bf16:
        ap = T.round_bf16(ap, True, True)
    b2 = cr(ap, 64, (1, 1), "branch2_conv_1x1")
    b3 = cr(x, 64, (1, 1), "branch3_conv_1x1")
    b3a = cr(b3, 32, (3, 1), "branch3_conv_3x1")
    b3b = cr(b3, 32, (1, 3), "branch3_conv_1x3")
    b4 = cr(x, 64, (1, 1), "branch4_conv_1x1")
    b4 = cr(b4, 64, (3, 3), "branch4_conv_3x3")
    b4a = cr(b4, 32, (3, 1), "branch4_conv_1x3")      # (sic) names swapped in the reference
    b4b = cr(b4, 32, (1, 3), "branch4_conv_3x1")
    hyper = torch.cat([b1, b2, b3a, b3b, b4a, b4b], dim=-1)
    out = cr(hyper, c, (1, 1), "residual_conv") + x
    return T.round_bf16(out, True, True) if P.emulate_bf16 else out


def deform_conv_2d(P, x, num_outputs, name, dg=4, no_bias=False, relu=False):
    """custom_op.deform_conv_2d — utility/custom_op.py:128-146 (channels_last: transposes around the op).
    Offset conv: tf.layers.conv2d default name 'conv2d', zero-init kernel+bias; kernel var OIHW.
    relu: the activation the graphs apply right after (net/danet_deform.py:281), folded in here so that the bf16-storage
    emulation can round where the MI355X build stores (sampled columns, the activated output, their gradients)."""
    off = conv(P, x, 2 * dg * 9, (3, 3), 1, name + "/conv2d", relu=False, init="zeros")
    if P.impose is not None and name + "/conv2d/kernel" in P.impose.get("offsets", {}):
        # the sampling cell floor(position) is a discrete decision too: take the offsets' VALUES from the recorded run (straight-through:
        # the gradient still flows into this graph's offset convolution)
        ref = P.impose["offsets"][name + "/conv2d/kernel"]
        assert ref.shape == off.shape, (name, tuple(ref.shape), tuple(off.shape))
        off = off + (ref - off).detach()
    w = P.get(name + "/kernel", (num_outputs, x.shape[-1], 3, 3), "glorot_oihw")
    b = None if no_bias else P.get(name + "/bias", (num_outputs,), "zeros")
    xn, on = x.permute(0, 3, 1, 2), off.permute(0, 3, 1, 2)
    if not P.emulate_bf16:
        y = deform_conv_forward(xn, w, on, 1, 1, dg).permute(0, 2, 3, 1)
        y = y if b is None else y + b
        return _relu(P, y, name) if relu else y
    w = T.round_bf16(w, True, False)
    col = T.round_bf16(deform_im2col(xn, on, 3, 3, 1, 1, dg), True, True)       # the sampled operand is a 16-bit MFMA input
    B, C, K, Ho, Wo = col.shape
    y = torch.einsum("ok,bkn->bon", w.reshape(num_outputs, C * K), col.reshape(B, C * K, Ho * Wo)).reshape(B, num_outputs, Ho, Wo)
    y = y.permute(0, 2, 3, 1)
    y = y if b is None else y + b
    y = T.round_bf16(y, False, True) if relu else y
    y = _relu(P, y, name) if relu else y
    return T.round_bf16(y, True, not relu)


def se_inception_block_v2(P, x, name):
    """DAN-Deform context module V2 — net/danet_deform.py:267-290."""
    c = x.shape[-1]
    d = conv(P, x, 256, (1, 1), 1, name + "/conv_1x1_down", relu=True)
    y = deform_conv_2d(P, d, 256, name + "/deform_conv", dg=4, no_bias=False, relu=True)
    out = conv(P, y, c, (1, 1), 1, name + "/conv_1x1_up", relu=True) + x
    return T.round_bf16(out, True, True) if P.emulate_bf16 else out


def features_conv_only(P, feats, stage1=None, name=None):
    """net/danet_deform.py:328-366: *_conv_only variants — (stage 2: the C//3 + (C - C//3) input mix, then) one deformable 3x3 + ReLU."""
    outs = []
    for i, f in enumerate(feats):
        c = f.shape[-1]
        if stage1 is None:
            nm = "{}/predict_stage1_conv{}".format(name or "prediction_modules_stage1", i)
        else:
            n2 = name or "prediction_modules_stage2"
            s1 = conv(P, stage1[i].detach(), c // 3, (1, 1), 1, "{}/satge1_conv_1x1_{}".format(n2, i), relu=True)
            rs = conv(P, f, c - c // 3, (1, 1), 1, "{}/residual_conv_1x1_{}".format(n2, i), relu=True)
            f = torch.cat([s1, rs], dim=-1)
            nm = "{}/predict_stage2_conv{}".format(n2, i)
        outs.append(deform_conv_2d(P, f, c, nm, dg=4, no_bias=False, relu=True))
    return outs


def get_features_stage1(P, feats, block, name="prediction_modules_stage1"):
    """net/danet.py:920-929 / net/danet_deform.py:292-301."""
    return [block(P, f, "{}/predict_stage1_{}".format(name, i)) for i, f in enumerate(feats)]


def get_features_stage2(P, stage1, feats, block, name="prediction_modules_stage2"):
    """net/danet.py:931-954: stop_gradient(stage1) -> 1x1 C//3; backbone feat -> 1x1 C - C//3; concat; block."""
    outs = []
    for i, f in enumerate(feats):
        c = f.shape[-1]
        s1 = conv(P, stage1[i].detach(), c // 3, (1, 1), 1, "{}/satge1_conv_1x1_{}".format(name, i), relu=True)
        rs = conv(P, f, c - c // 3, (1, 1), 1, "{}/residual_conv_1x1_{}".format(name, i), relu=True)
        outs.append(block(P, torch.cat([s1, rs], dim=-1), "{}/predict_stage2_{}".format(name, i)))
    return outs


# ------------------------------------------------------------------ ResNet backbone (net/resnet_danet.py:92-228), unused by the scripts
RESNET_UNITS = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}


def _bn(P, x, scope, training, eps=1e-5):
    c = x.shape[-1]
    gamma, beta = P.get(scope + "/bn/gamma", (c,), 1.0), P.get(scope + "/bn/beta", (c,), "zeros")
    mm, mv = P.get(scope + "/bn/moving_mean", (c,), "zeros"), P.get(scope + "/bn/moving_variance", (c,), 1.0)
    y = T.batch_norm_train(x, gamma, beta, eps)[0] if training else T.batch_norm_infer(x, gamma, beta, mm, mv, eps)
    return T.round_bf16(y, True, True) if P.emulate_bf16 else y


def conv_bn(P, x, filters, ksize, stride, scope, training, padding="same", relu=False):
    """conv_bn / conv_bn_relu (net/resnet_danet.py:176-218): conv without bias -> batch_normalization -> (relu)."""
    w = P.get(scope + "/conv2d/kernel", (ksize[0], ksize[1], x.shape[-1], filters), "glorot")
    if P.emulate_bf16:
        w = T.round_bf16(w, True, False)
    y = (T.conv2d_valid if padding == "valid" else T.conv2d_same)(x, w, None, stride=stride)
    if P.emulate_bf16:
        y = T.round_bf16(y, True, True)
    y = _bn(P, y, scope, training)
    return torch.relu(y) if relu else y


def bottleneck_block(P, x, filters, scope, training, need_reduce=True, is_root=False):
    """net/resnet_danet.py:157-173."""
    s = 1 if (not need_reduce) or is_root else 2
    shortcut = conv_bn(P, x, filters * 2, (1, 1), s, scope + "/shortcut", training, "valid") if need_reduce else x
    y = conv_bn(P, x, filters // 2, (1, 1), s, scope + "/reduce", training, "valid", relu=True)
    y = conv_bn(P, F.pad(y, (0, 0, 1, 1, 1, 1)), filters // 2, (3, 3), 1, scope + "/block_3x3", training, "valid", relu=True)
    y = conv_bn(P, y, filters * 2, (1, 1), 1, scope + "/increase", training)
    out = torch.relu(y + shortcut)
    return T.round_bf16(out, True, True) if P.emulate_bf16 else out


def resnet_get_featmaps(P, x, depth=50, training=False):
    """ResNetBackbone.get_featmaps (net/resnet_danet.py:114-155; freeze=False): 7x7/2 stem on the explicitly padded image, 3x3/2
    max-pool, four bottleneck stages, two extra stride-2 stages -> six maps of 256/512/1024/2048/512/256 channels."""
    input_depth = [128, 256, 512, 1024]
    y = conv_bn(P, F.pad(x, (0, 0, 3, 3, 3, 3)), input_depth[0] // 2, (7, 7), 2, "block_0/conv_1", training, "valid", relu=True)
    y = T.max_pool_3x3_s2_same(y)
    feats, is_root = [], True
    for ind, n_units in enumerate(RESNET_UNITS[depth]):
        need_reduce = True
        for u in range(1, n_units + 1):
            y = bottleneck_block(P, y, input_depth[ind], "block_{}/conv_{}".format(ind + 1, u), training, need_reduce, is_root)
            need_reduce, is_root = False, False
        feats.append(y)
    y = conv_bn(P, y, 512, (1, 1), 1, "additional_layers/conv6_1", training, relu=True)
    y = conv_bn(P, y, 512, (3, 3), 2, "additional_layers/conv6_2", training, relu=True)
    feats.append(y)
    y = conv_bn(P, y, 128, (1, 1), 1, "additional_layers/conv7_1", training, relu=True)
    y = conv_bn(P, y, 256, (3, 3), 2, "additional_layers/conv7_2", training, relu=True)
    feats.append(y)
    return feats


def flatten_preds(preds, k):
    """reshape_pred — train_dan.py:340-355 / train_sfd.py:293-304: [B,H,W,A*k] -> [B, HW*A, k], concat levels."""
    b = preds[0].shape[0]
    return torch.cat([p.reshape(b, -1, k) for p in preds], dim=1)


# ------------------------------------------------------------------ whole-model forwards
def sfd_forward(P, x):
    """train_sfd.py:286-304 / eval_sfd.py:262-283: returns (loc [B,A,4], cls [B,A,2])."""
    feats = get_featmaps(P, x)
    loc, cls = predict_module(P, feats, [1] * 6, [3] + [1] * 5, [1] * 6, "multibox_head")
    return flatten_preds(loc, 4), flatten_preds(cls, 2)


def pb_forward(P, x):
    """train_pb.py:396-420: face head on all 6 levels, head head on levels 1.., body head on levels 2..."""
    feats = get_featmaps(P, x)
    feats = build_lfpn(P, feats, skip_last=3, name="lfpn")
    feats = context_pred_module(P, feats)
    fl, fc = predict_module(P, feats, [1] + [3] * 5, [3] + [1] * 5, [1] * 6, "predict_face")
    hl, hc = predict_module(P, feats[1:], [1] * 5, [1] * 5, [1] * 5, "predict_head")
    bl, bc = predict_module(P, feats[2:], [1] * 4, [1] * 4, [1] * 4, "predict_body")
    return {"face": (flatten_preds(fl, 4), flatten_preds(fc, 2)),
            "head": (flatten_preds(hl, 4), flatten_preds(hc, 2)),
            "body": (flatten_preds(bl, 4), flatten_preds(bc, 2))}


def dan_forward(P, x, deform=False):
    """train_dan.py:410-428 / eval_dan.py:316-371: two-stage heads; returns stage-1 and stage-2 (loc, cls)."""
    block = se_inception_block_v2 if deform else se_inception_block_v1
    feats = get_featmaps(P, x)
    feats = build_lfpn(P, feats, skip_last=3, name="lfpn", fused_channels=256)
    s1 = get_features_stage1(P, feats, block)
    s1p = build_lfpn(P, s1, skip_last=3, name="lfpn_stage1", fused_channels=256)
    l1, c1 = predict_module(P, s1p, [1] * 6, [1] * 6, [1] * 6, "predict_face", shared_conv=True)
    s2 = get_features_stage2(P, s1p, feats, block)   # stage 2 consumes the LFPN'd stage-1 maps (train_dan.py:423)
    s2p = build_lfpn(P, s2, skip_last=3, name="lfpn_stage2", fused_channels=256)
    l2, c2 = predict_module(P, s2p, [1] * 6, [3] + [1] * 5, [1] * 6, "predict_cascade", shared_conv=True)
    return (flatten_preds(l1, 4), flatten_preds(c1, 2)), (flatten_preds(l2, 4), flatten_preds(c2, 2))


def preprocess_synthetic(img_u8_rgb):
    """preprocess_for_eval arithmetic — preprocessing/dan_preprocessing.py:55-57,755-758:
    float - (R,G,B means), then RGB->BGR, no scaling.  img [B,H,W,3] uint8 RGB -> float32 BGR."""
    means = torch.tensor([123.68, 116.78, 103.94])
    x = img_u8_rgb.to(torch.float32) - means
    return x.flip(-1).contiguous()
