"""S3FD graph on the libdanhip kernels — mirrors the operator surface of the reference's net/sfd_net.py
(class VGG16Backbone: l2_normalize, conv_relu, conv_block, get_featmaps, multibox_head; same argument meaning),
native layout NHWC bf16 activations / fp32 HWIO variables under the reference's TF variable names.
"""
import os

import torch

from .. import ops
from .variables import VariableStore


# inference: conv1_2 / conv2_2 feed nothing but their pool, so their full-resolution maps are never written (DANHIP_POOL_ONLY=0: A/B)
POOL_ONLY = os.environ.get("DANHIP_POOL_ONLY", "1") == "1"


class VGG16Backbone(object):
    def __init__(self, data_format="channels_last", bn_epsilon=1e-5, bn_momentum=0.997, use_fused_bn=True, variables=None):
        if data_format != "channels_last":
            raise ValueError("the MI355X build is NHWC-native: use data_format='channels_last' "
                             "(the reference's eval scripts' default, eval_sfd.py:63)")
        self._data_format = data_format
        self._bn_epsilon = bn_epsilon
        self._bn_momentum = bn_momentum
        self._use_fused_bn = use_fused_bn
        self.vs = variables if variables is not None else VariableStore()

    # ---- layers ---------------------------------------------------------------------------------------------
    def l2_normalize(self, inputs, init_value, training, name=None):
        """net/sfd_net.py:68-79."""
        w = self.vs.get((name or "l2_normalize") + "/weight", (inputs.shape[-1],), init_value)
        return ops.l2_normalize(inputs, w)

    def conv2d(self, inputs, filters, kernel_size, strides, scope, relu, out_f32=False, residual=None, init="glorot", pool=False, pool_only=False):
        kh, kw = kernel_size
        s = strides[0] if isinstance(strides, (tuple, list)) else strides
        cin = inputs.shape[-1]
        w = self.vs.get(scope + "/kernel", (kh, kw, getattr(inputs, "_real_channels", cin), filters), init)
        b = self.vs.get(scope + "/bias", (filters,), "zeros")
        return ops.conv2d(inputs, w, b, stride=s, relu=relu, out_f32=out_f32, residual=residual, pool=pool, pool_only=pool_only)

    def conv_relu(self, inputs, filters, kernel_size, strides, scope, padding="same", dilate_rate=1, reuse=None, pool=False, pool_only=False):
        """net/sfd_net.py:81-89.  pool=True (not in the reference signature): the caller max-pools the result next, so the conv
        kernel produces the pooled map in its epilogue (ops.max_pool_2x2 then just hands it over)."""
        assert padding == "same" and dilate_rate == 1
        return self.conv2d(inputs, filters, kernel_size, strides, scope + "/conv2d", relu=True, pool=pool, pool_only=pool_only)

    # ---- batch-norm surface (net/sfd_net.py:91-119): defined on every VGG16Backbone, used by no VGG graph ------------
    def _bn(self, inputs, scope, training, relu):
        c = inputs.shape[-1]
        gamma = self.vs.get(scope + "/bn/gamma", (c,), 1.0)
        beta = self.vs.get(scope + "/bn/beta", (c,), "zeros")
        mm = self.vs.buffer(scope + "/bn/moving_mean", (c,), 0.0)
        mv = self.vs.buffer(scope + "/bn/moving_variance", (c,), 1.0)
        if training:
            return ops.batch_norm_train(inputs, gamma, beta, mm, mv, eps=self._bn_epsilon, momentum=self._bn_momentum, relu=relu)
        return ops.batch_norm_infer(inputs, gamma, beta, mm, mv, eps=self._bn_epsilon, relu=relu)

    def _conv_nobias(self, inputs, filters, kernel_size, strides, scope):
        kh, kw = kernel_size
        s = strides[0] if isinstance(strides, (tuple, list)) else strides
        w = self.vs.get(scope + "/conv2d/kernel", (kh, kw, inputs.shape[-1], filters), "glorot")
        return ops.conv2d(inputs, w, None, stride=s, relu=False)

    def conv_bn_relu(self, inputs, filters, kernel_size, strides, scope, training, padding="same", dilate_rate=1, reuse=None):
        """net/sfd_net.py:91-101: conv (no bias) -> batch_normalization(momentum 0.997, eps 1e-5) -> relu."""
        assert padding == "same" and dilate_rate == 1
        return self._bn(self._conv_nobias(inputs, filters, kernel_size, strides, scope), scope, training, relu=True)

    def bn_relu(self, inputs, scope, training, reuse=None):
        """net/sfd_net.py:103-107."""
        return self._bn(inputs, scope, training, relu=True)

    def conv_bn(self, inputs, filters, kernel_size, strides, scope, training, padding="same", dilate_rate=1, reuse=None):
        """net/sfd_net.py:109-119."""
        assert padding == "same" and dilate_rate == 1
        return self._bn(self._conv_nobias(inputs, filters, kernel_size, strides, scope), scope, training, relu=False)

    def conv_block(self, inputs, num_blocks, filters, kernel_size, strides, name, reuse=None, pool_after=False, pool_only=False):
        """net/sfd_net.py:121-125.  pool_after: the block is followed by max_pooling2d (get_featmaps): fuse it into the last conv.
        pool_only: nothing but that pool reads the block's output (conv1, conv2) - at inference its full-resolution map is never written."""
        for ind in range(1, num_blocks + 1):
            last = ind == num_blocks
            inputs = self.conv_relu(inputs, filters, kernel_size, strides, "{0}/{0}_{1}".format(name, ind), pool=pool_after and last,
                                    pool_only=pool_only and pool_after and last)
        return inputs

    def get_featmaps(self, inputs, training=False):
        """net/sfd_net.py:127-156.  inputs: bf16 NHWC BGR mean-subtracted, channels zero-padded to 8
        (ops.preprocess_u8 makes it; `_real_channels` = 3 keeps the TF kernel shape [3,3,3,64])."""
        feature_layers = []
        # precision "mixed" (inference, SFDModel.precision): 16-bit backbone, but every tap that feeds a detection head is widened to fp32
        # first, so the L2 normalisation and the head convolutions run on the fp32 kernels (how much of the 16-bit
        # path's box error the heads cause - DESIGN section 4 has the measurement)
        tap = (lambda t: t.float()) if getattr(self, "fp32_heads", False) and not torch.is_grad_enabled() else (lambda t: t)
        inputs = self.conv_block(inputs, 2, 64, (3, 3), (1, 1), "conv1", pool_after=True, pool_only=POOL_ONLY)
        inputs = ops.max_pool_2x2(inputs)
        inputs = self.conv_block(inputs, 2, 128, (3, 3), (1, 1), "conv2", pool_after=True, pool_only=POOL_ONLY)
        inputs = ops.max_pool_2x2(inputs)
        inputs = self.conv_block(inputs, 3, 256, (3, 3), (1, 1), "conv3", pool_after=True)
        feature_layers.append(self.l2_normalize(tap(inputs), 10, training, "l2_norm_layer_3"))
        inputs = ops.max_pool_2x2(inputs)
        inputs = self.conv_block(inputs, 3, 512, (3, 3), (1, 1), "conv4", pool_after=True)
        feature_layers.append(self.l2_normalize(tap(inputs), 8, training, "l2_norm_layer_4"))
        inputs = ops.max_pool_2x2(inputs)
        inputs = self.conv_block(inputs, 3, 512, (3, 3), (1, 1), "conv5", pool_after=True)
        feature_layers.append(self.l2_normalize(tap(inputs), 5, training, "l2_norm_layer_5"))
        inputs = ops.max_pool_2x2(inputs)
        inputs = self.conv_relu(inputs, 1024, (3, 3), (1, 1), "fc6")
        inputs = self.conv_relu(inputs, 1024, (1, 1), (1, 1), "fc7")
        feature_layers.append(tap(inputs))
        inputs = self.conv_relu(inputs, 256, (1, 1), (1, 1), "additional_layers/conv6_1")
        inputs = self.conv_relu(inputs, 512, (3, 3), (2, 2), "additional_layers/conv6_2")
        feature_layers.append(tap(inputs))
        inputs = self.conv_relu(inputs, 128, (1, 1), (1, 1), "additional_layers/conv7_1")
        inputs = self.conv_relu(inputs, 256, (3, 3), (2, 2), "additional_layers/conv7_2")
        feature_layers.append(tap(inputs))
        return feature_layers

    def predict_heads(self, feature_layers, pos_maxout, neg_maxout, num_anchors_depth_per_layer, name, shared_conv=False):
        """Shared body of multibox_head (net/sfd_net.py:159-219), pb_net.get_predict_module (:230-290) and
        danet.get_predict_module (:469-532).  loc_i and cls_i stay separate TF variables but run as ONE 3x3 conv
        (their HWIO kernels concatenated on Cout: the feature map is read once — HBM-bound head), then
        max-out + reshape_pred write straight into the level-concatenated [B, A, 4] / [B, A, 2] buffers."""
        B = feature_layers[0].shape[0]
        assert all(d == 1 for d in num_anchors_depth_per_layer), "one anchor per cell (all reference configs)"
        A = sum(f.shape[1] * f.shape[2] for f in feature_layers)
        dev = feature_layers[0].device
        loc = torch.zeros((B, A, 4), dtype=torch.float32, device=dev)
        cls = torch.zeros((B, A, 2), dtype=torch.float32, device=dev)
        off = 0
        for ind, feat in enumerate(feature_layers):
            if shared_conv:
                feat = self.conv_relu(feat, feat.shape[-1], (3, 3), (1, 1), "{}/shared_conv_{}".format(name, ind))
            c = feat.shape[-1]
            ncls = pos_maxout[ind] + neg_maxout[ind]
            wl = self.vs.get("{}/loc_{}/kernel".format(name, ind), (3, 3, c, 4), "glorot")
            bl = self.vs.get("{}/loc_{}/bias".format(name, ind), (4,), "zeros")
            wc = self.vs.get("{}/cls_{}/kernel".format(name, ind), (3, 3, c, ncls), "glorot")
            bc = self.vs.get("{}/cls_{}/bias".format(name, ind), (ncls,), "zeros")
            pre = "{}/loc_{}/".format(name, ind), "{}/cls_{}/".format(name, ind)
            wf = self.vs.fuse((pre[0] + "kernel", pre[1] + "kernel"), axis=3)          # one block in the trainer's flat buffer, or None
            bf = self.vs.fuse((pre[0] + "bias", pre[1] + "bias"), axis=0)
            if wf is None or bf is None:
                wf, bf = torch.cat([wl, wc], dim=3).contiguous(), torch.cat([bl, bc])
            h = ops.conv2d(feat, wf, bf, stride=1, relu=False, out_f32=True)
            if ops.TRACE is not None and ncls > 2:           # the max-out decision of this level (tests: imposed on the oracle graph)
                ops.TRACE.setdefault("maxout", {})[id(wc)] = h.detach()[..., 4:]
            loc, cls = ops.head_split(h, loc, cls, neg_maxout[ind], pos_maxout[ind], off)
            off += feat.shape[1] * feat.shape[2]
        return loc, cls

    def multibox_head(self, feature_layers, pos_maxout, neg_maxout, num_anchors_depth_per_layer):
        """net/sfd_net.py:159-219 + reshape/concat of train_sfd.py:293-304: returns (location_pred [B,A,4],
        cls_pred [B,A,2]) already in (level, y, x, anchor) order."""
        return self.predict_heads(feature_layers, pos_maxout, neg_maxout, num_anchors_depth_per_layer, "multibox_head")


import contextlib


@contextlib.contextmanager
def precision_scope(precision):
    """precision "split": the fp32 inference path with its convolutions as split-operand products on the fp16 MFMA (ops.SPLIT_EVAL,
    csrc/split_infer.hip) for the duration of the forward pass; anything else leaves the ops' context alone."""
    ctx = ops.context()
    prev = ctx.SPLIT_EVAL
    ctx.SPLIT_EVAL = precision == "split"
    try:
        yield
    finally:
        ctx.SPLIT_EVAL = prev


def prepare_input(img_u8_rgb, precision="act"):
    """uint8 RGB [B,H,W,3] (device) -> network input.  precision "act": the build's 16-bit activation type (training and fast inference);
    "fp32": the fp32 inference path (ops._f32_infer: every layer then runs the fp32 kernels); "split": fp32 maps between the ops as in
    "fp32", convolutions as three-limb-product half convolutions (inside precision_scope)."""
    if precision in ("fp32", "split"):
        return ops.preprocess_f32(img_u8_rgb)
    x = ops.preprocess_u8(img_u8_rgb)
    x._real_channels = 3
    return x
