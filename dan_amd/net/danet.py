"""DAN graph on the libdanhip kernels — mirrors net/danet.py (class VGG16Backbone: backbone, build_lfpn :339-380 with a
256-channel fused conv, se_inception_block V1 :842-918, get_features_stage1 :920-929, get_features_stage2 :931-954,
get_predict_module with shared conv :469-532) with the reference's variable names."""
import os

import torch

from .. import ops
from . import pb_net


class VGG16Backbone(pb_net.VGG16Backbone):
    def build_lfpn(self, feature_layers, skip_last=3, name=None):
        """net/danet.py:339-380: as PyramidBox's LFPN but the fused 3x3 conv always has 256 filters."""
        return super().build_lfpn(feature_layers, skip_last, name, fused_channels=256)

    def build_bi_lfpn(self, feature_layers, skip_last=3, name=None):
        """net/danet.py:191-249 (defined on the backbone, used by no script): EVERY level but the last is merged with its
        coarser neighbour — 1x1 'lateral' conv on the coarser map, bilinear resize to this level, add, 3x3 'fused_conv' to 256
        channels; no activation anywhere; the last level passes through."""
        name = name or "lfpn"
        outs = []
        for ind, featmap in enumerate(feature_layers[:-1]):
            sc = "{}/fpn_{}".format(name, ind)
            up = self.conv2d(feature_layers[ind + 1], featmap.shape[-1], (1, 1), 1, sc + "/lateral", relu=False)
            merged = ops.resize_bilinear_add(up, featmap)                      # featmap + resize_bilinear(up, size(featmap))
            outs.append(self.conv2d(merged, 256, (3, 3), 1, sc + "/fused_conv", relu=False))
        return outs + [feature_layers[-1]]

    def build_reverse_lfpn(self, feature_layers, skip_last=3, name=None):
        """net/danet.py:382-412 (defined on the backbone, used by no script): the bottom-up counterpart of build_lfpn — for ind = 0 ..
        skip_last-1 a 3x3 / stride-2 'downsample_conv' of the running map (the first level to start with) to the next level's channels,
        a 1x1 'lateral' on that next level, their SUM carried on, and a 3x3 'fused_conv' to 256 channels as the level's output; no
        activation anywhere; level 0 and the levels beyond skip_last pass through."""
        name = name or "reverse_lfpn"
        outs = []
        down = None
        for ind in range(0, skip_last):
            sc = "{}/reverse_fpn_{}".format(name, ind)
            down_channels = feature_layers[ind + 1].shape[-1]
            if down is None:
                down = feature_layers[ind]
            down = self.conv2d(down, down_channels, (3, 3), 2, sc + "/downsample_conv", relu=False)
            lateral = self.conv2d(feature_layers[ind + 1], down_channels, (1, 1), 1, sc + "/lateral", relu=False)
            down = ops.add(lateral, down)
            outs.append(self.conv2d(down, 256, (3, 3), 1, sc + "/fused_conv", relu=False))
        return [feature_layers[0]] + outs + list(feature_layers[skip_last + 1:])

    def _cr(self, inputs, filters, ksize, name):
        return self.conv2d(inputs, filters, ksize, 1, name, relu=True)

    def _residual_conv(self, hyper, filters, name, x):
        """relu(conv1x1(hyper)) + x (net/danet.py:913-918): fused into the conv epilogue when no gradient is tracked; otherwise a
        separate add whose backward hands dY to both producers (the fused form would need the pre-residual ReLU mask in backward)."""
        if not torch.is_grad_enabled():
            return self.conv2d(hyper, filters, (1, 1), 1, name, relu=True, residual=x)
        return ops.add(self._cr(hyper, filters, (1, 1), name), x)

    # False (or DANHIP_FUSED_CONTEXT=0 at import): the block as ten separate convolutions + concat + add (the round-3 form; A/B and tests).
    # Set it BEFORE a trainer is built: FlatParams lays the block's kernels out as strided views that only the fused call consumes
    # (ops.packed_weights raises on the unfused path otherwise).
    FUSED_CONTEXT_BLOCK = os.environ.get("DANHIP_FUSED_CONTEXT", "1") == "1"

    def se_inception_block(self, inputs, name=None):
        """DAN context module V1 — net/danet.py:842-918."""
        c = inputs.shape[-1]
        # (training AND 16-bit inference: once the trainer has laid the three fused 1x1 kernels out side by side they are strided views, which
        # only the fused call consumes; the fp32 inference path reads the TF variables as they are)
        if self.FUSED_CONTEXT_BLOCK and inputs.dtype == ops.ACT and c % 64 == 0 and inputs.is_contiguous():
            return self._se_inception_block_fused(inputs, name)
        b1 = self._cr(inputs, 64, (1, 1), name + "/branch1_conv_1x1")
        b2 = self._cr(ops.avg_pool_2x2_s1(inputs), 64, (1, 1), name + "/branch2_conv_1x1")
        b3 = self._cr(inputs, 64, (1, 1), name + "/branch3_conv_1x1")
        b3a = self._cr(b3, 32, (3, 1), name + "/branch3_conv_3x1")
        b3b = self._cr(b3, 32, (1, 3), name + "/branch3_conv_1x3")
        b4 = self._cr(inputs, 64, (1, 1), name + "/branch4_conv_1x1")
        b4 = self._cr(b4, 64, (3, 3), name + "/branch4_conv_3x3")
        b4a = self._cr(b4, 32, (3, 1), name + "/branch4_conv_1x3")       # (sic) the reference swaps these two names
        b4b = self._cr(b4, 32, (1, 3), name + "/branch4_conv_3x1")
        hyper = ops.concat([b1, b2, b3a, b3b, b4a, b4b])
        return self._residual_conv(hyper, c, name + "/residual_conv", inputs)

    def _se_inception_block_fused(self, inputs, name):
        """The same block as ONE autograd node over channel-slice views (ops._ContextBlock): the three 1x1s of branches 3, 4 and 2 run as
        one convolution (their kernels side by side in the trainer's flat buffer: VariableStore.fuse), branch 2's average pool moves behind
        its 1x1, each 3x1 / 1x3 pair runs as one 3x3 ("plus" kernel), every branch writes its slice of the concat buffer directly and reads
        its slice of the concat's gradient in place.
        Variables are created in the reference's order under the reference's names."""
        c = inputs.shape[-1]
        V = self.vs

        def var(scope, kh, kw, cin, cout):
            return (V.get(name + "/" + scope + "/kernel", (kh, kw, cin, cout), "glorot"), V.get(name + "/" + scope + "/bias", (cout,), "zeros"))

        w1, c1 = var("branch1_conv_1x1", 1, 1, c, 64)
        w2, c2 = var("branch2_conv_1x1", 1, 1, c, 64)
        w3, c3 = var("branch3_conv_1x1", 1, 1, c, 64)
        w3a, c3a = var("branch3_conv_3x1", 3, 1, 64, 32)
        w3b, c3b = var("branch3_conv_1x3", 1, 3, 64, 32)
        w4, c4 = var("branch4_conv_1x1", 1, 1, c, 64)
        w43, c43 = var("branch4_conv_3x3", 3, 3, 64, 64)
        w4a, c4a = var("branch4_conv_1x3", 3, 1, 64, 32)           # (sic) the reference swaps these two names
        w4b, c4b = var("branch4_conv_3x1", 1, 3, 64, 32)
        wr, cr = var("residual_conv", 1, 1, 256, c)
        pre = name + "/"
        kn = (pre + "branch3_conv_1x1/kernel", pre + "branch4_conv_1x1/kernel", pre + "branch2_conv_1x1/kernel")
        bn = (pre + "branch3_conv_1x1/bias", pre + "branch4_conv_1x1/bias", pre + "branch2_conv_1x1/bias")
        wcat, ccat = V.fuse(kn, axis=3), V.fuse(bn, axis=0)        # one block each in the trainer's flat buffer, or None (plain autograd)
        if wcat is None or ccat is None:
            wcat, ccat = V.build(kn, 3), V.build(bn, 0)

        def plus(va, vb):
            """The 3x1 and the 1x3 convolution of one input as ONE 3x3 kernel 64 -> 64: 3x1 taps in the middle column for outputs 0..31,
            1x3 taps in the middle row for outputs 32..63 (a "plus" block of the flat buffers, or built by torch ops for plain autograd)."""
            kk, bk = (pre + va + "/kernel", pre + vb + "/kernel"), (pre + va + "/bias", pre + vb + "/bias")
            wp, cp = V.fuse(kk, axis="plus"), V.fuse(bk, axis=0)
            if wp is None or cp is None:
                wp, cp = V.build(kk, "plus"), V.build(bk, 0)
            return wp, cp

        p3 = plus("branch3_conv_3x1", "branch3_conv_1x3")
        p4 = plus("branch4_conv_1x3", "branch4_conv_3x1")
        params = [(w1, c1), (wcat, ccat), p3, (w43, c43), p4, (wr, cr)]
        # gradient buckets: descending position in the flat buffer (the fused block stands where branch3_conv_1x1 stood)
        hooks = [wr, w4b, w4a, w43, w3b, w3a, w4, w3, w2, w1]
        trace = {"b1": w1, "b2": w2, "b3": w3, "b3a": w3a, "b3b": w3b, "b4": w4, "b43": w43, "b4a": w4a, "b4b": w4b, "res": wr}
        return ops.context_block(inputs, params, hooks, trace)

    def get_features_stage1(self, feature_layers, name=None):
        """net/danet.py:920-929."""
        name = name or "prediction_modules_stage1"
        return [self.se_inception_block(f, "{}/predict_stage1_{}".format(name, i)) for i, f in enumerate(feature_layers)]

    def get_features_stage2(self, feature_stage1, feature_layers, name=None):
        """net/danet.py:931-954: stop_gradient(stage1) -> 1x1 (C//3, ReLU); backbone feature -> 1x1 (C - C//3, ReLU);
        concat; context block."""
        name = name or "prediction_modules_stage2"
        outs = []
        for i, f in enumerate(feature_layers):
            c = f.shape[-1]
            s1n, rsn = "{}/satge1_conv_1x1_{}".format(name, i), "{}/residual_conv_1x1_{}".format(name, i)      # (sic)
            if self.FUSED_STAGE2_MIX and f.dtype == ops.ACT and c % 64 == 0 and f.is_contiguous() and feature_stage1[i].shape == f.shape:
                mixed = self._stage2_mix_fused(feature_stage1[i], f, s1n, rsn)
            else:
                s1 = self._cr(ops.stop_gradient(feature_stage1[i]), c // 3, (1, 1), s1n)
                rs = self._cr(f, c - c // 3, (1, 1), rsn)
                mixed = ops.concat([s1, rs])
            outs.append(self.se_inception_block(mixed, "{}/predict_stage2_{}".format(name, i)))
        return outs

    # False (or DANHIP_FUSED_STAGE2_MIX=0 at import): the two ragged 1x1 convolutions + concat (the form up to round 4; A/B and tests); as
    # FUSED_CONTEXT_BLOCK, a choice to make before the trainer exists
    FUSED_STAGE2_MIX = os.environ.get("DANHIP_FUSED_STAGE2_MIX", "1") == "1"

    def _stage2_mix_fused(self, stage1, f, s1n, rsn):
        """concat([relu(conv1x1(stop_gradient(stage1), C // 3)), relu(conv1x1(f, C - C // 3))]) as ONE 1x1 over the (never written) channel
        concatenation [stage1 | f] with the block-diagonal kernel diag(W1, W2) - ops._ConcatMix.  The two TF variables keep their names and
        shapes: in the trainer's flat buffer they are the diagonal blocks of one [1, 1, 2C, C] block (VariableStore.fuse "blockdiag")."""
        c = f.shape[-1]
        c3 = c // 3
        V = self.vs
        w1 = V.get(s1n + "/kernel", (1, 1, c, c3), "glorot")
        b1 = V.get(s1n + "/bias", (c3,), "zeros")
        w2 = V.get(rsn + "/kernel", (1, 1, c, c - c3), "glorot")
        b2 = V.get(rsn + "/bias", (c - c3,), "zeros")
        kk, bk = (s1n + "/kernel", rsn + "/kernel"), (s1n + "/bias", rsn + "/bias")
        wv, bv = V.fuse(kk, axis="blockdiag"), V.fuse(bk, axis=0)
        if wv is None or bv is None:                   # no trainer laid the block out: build it (plain autograd slices its gradient apart again)
            wv, bv = V.build(kk, "blockdiag"), V.build(bk, 0)
        return ops.concat_conv1x1_relu(stage1, f, wv, bv, split=(c, c3), trace_params=(w1, w2))

    def get_predict_module(self, feature_layers, pos_maxout, neg_maxout, num_anchors_depth_per_layer, name=None):
        """net/danet.py:469-532: shared 3x3 conv (ReLU) in front of the loc / cls convs of every level."""
        return self.predict_heads(feature_layers, pos_maxout, neg_maxout, num_anchors_depth_per_layer, name or "predict_face", shared_conv=True)
