"""Host-side mirror of the reference's utility/custom_op.py (the TF custom-op plugin boundary): the same callables
— small_mining_match (:48), dynamic_anchor_routing (:52), deform_conv_op / deform_conv_2d (:62,:128-146) — bound to
libdanhip instead of tf.load_op_library.  Argument order and meaning follow the TF op definitions
(cpp/ExtraLib/small_mining_match.cc:31-54, cpp/ExtraLib/dynamic_anchor_routing.cc:32-65, cpp/Deform/deform_conv.cc:51-167).
"""
import torch

from .. import _lib
from .._lib import DanhipError, call, ptr, stream
from .anchor_manipulator import small_mining_match  # noqa: F401  (re-exported under the reference's module path)


def dynamic_anchor_routing(anchors, gt_targets, labels, mask_in, feat_height, feat_width, anchor_depth, feat_strides, img_height, img_width,
                           trainging, thres, ignore_thres, seed=0, counter0=0, counter_dev=None):
    """custom_op.dynamic_anchor_routing (utility/custom_op.py:52).  Tensors may carry a leading batch dimension
    ([B,N,4] / [B,N]); img_height / img_width are accepted and ignored exactly as the reference kernel does.
    Training mode draws u(b,i) from the counter-based stream keyed by (seed, counter0) (the reference's
    std::random_device stream is not reproducible); counter_dev (int64 device tensor, optional) is added to counter0 on the device.
    Returns (mask_out int32, decode_out fp32) of the input's shape."""
    if not (0.0 <= thres < 1.0) or not (0.0 <= ignore_thres < 1.0):
        raise ValueError("thres / ignore_thres must be in [0, 1) (dynamic_anchor_routing.cc:526-530)")
    batched = labels.dim() == 2
    a = anchors.to(torch.float32).contiguous()
    g = gt_targets.to(torch.float32).contiguous()
    l = labels.to(torch.float32).contiguous()
    m = mask_in.to(torch.int32).contiguous()
    B = l.shape[0] if batched else 1
    N = l.shape[-1]
    if N != int(feat_height) * int(feat_width) * int(anchor_depth):
        raise ValueError("labels has %d rows, expected feat_height*feat_width*anchor_depth = %d" % (N, feat_height * feat_width * anchor_depth))
    mo = torch.empty(l.shape, dtype=torch.int32, device=l.device)
    do = torch.empty(l.shape + (4,), dtype=torch.float32, device=l.device)
    nws = _lib.lib().danhip_routing_workspace_bytes(N, B, 1 if trainging else 0)
    ws = torch.empty((nws + 7) // 8, dtype=torch.int64, device=l.device)
    if trainging:
        call("danhip_dynamic_anchor_routing_train", ptr(a), ptr(g), ptr(l), ptr(m), N, int(feat_height), int(feat_width), int(anchor_depth),
             int(feat_strides), B, float(thres), float(ignore_thres), int(seed), int(counter0), ptr(counter_dev), ptr(mo), ptr(do), ptr(ws), nws,
             stream())
    else:
        call("danhip_dynamic_anchor_routing_eval", ptr(a), ptr(g), ptr(l), ptr(m), N, int(feat_height), int(feat_width), int(anchor_depth),
             int(feat_strides), B, ptr(mo), ptr(do), ptr(ws), nws, stream())
    return mo, do


def deform_conv_op(x, filter, offset, rates, padding, strides, num_groups, deformable_group, bias=None, relu=False, data_format="NHWC"):
    """custom_op.deform_conv_op (utility/custom_op.py:62; DeformConvOp cpp/Deform/deform_conv.cc:51-167) on NHWC tensors:
    x bf16 [N,H,W,C], filter fp32 OIHW [Cout,C/num_groups,kh,kw] (the reference's variable layout, custom_op.py:134), offset bf16
    [N,Ho,Wo,dg*2*kh*kw].  rates / strides are the reference's 4-vectors [1,1,r,r] / [1,1,s,s].
    = deformable im2col (HIP gather kernel) followed by the MFMA convolution kernel run as a 1x1 GEMM over the samples,
    both inside danhip_deform_conv_fwd; the backward is danhip_deform_conv_bwd (DeformConvBackpropOp, :170-189).  `bias` / `relu` fuse the Python-side bias add of
    deform_conv_2d (custom_op.py:145) and the activation that follows it into the GEMM epilogue.

    The op attributes no graph of the reference uses are exact compositions of the SAME / one-group / NHWC kernels (round 3):
      data_format="NCHW"  (the op's default, deform_conv.cc:61): x [N,C,H,W] / offset [N,dg*2*kh*kw,Ho,Wo] -> [N,Cout,Ho,Wo], transposed around the call;
      padding="VALID"     (stride 1): the VALID output (ho, wo) samples h = ho + i*r + off — what the SAME output (ho + pad, wo + pad) samples
                          (GetWindowedOutputSize uses the undilated kernel, :473-479) — so the offsets are embedded into a SAME-sized
                          tensor, the SAME op runs, and the VALID window is cropped out: same sample positions, same arithmetic;
      num_groups=G > 1    (:487-515: G contiguous channel groups, weight [G, Cout/G, C/G*kh*kw]): one call per group on its channel slice,
                          with the deformable groups that slice covers (needs G | dg or dg | G), outputs concatenated.
    fp32 / fp64 activations stay unsupported on the training path (16-bit storage; the fp32 inference path is ops.*_f32)."""
    from .. import ops
    x, offset = ops._f32_in(x, offset)               # (split-operand inference: limb views are widened here, before any slicing / transposing)
    if data_format == "NCHW":
        y = deform_conv_op(x.permute(0, 2, 3, 1).contiguous(), filter, offset.permute(0, 2, 3, 1).contiguous(), rates, padding, strides, num_groups,
                           deformable_group, bias=bias, relu=relu)
        return y.permute(0, 3, 1, 2).contiguous()
    if data_format != "NHWC":
        raise ValueError("data_format must be 'NHWC' or 'NCHW' (deform_conv.cc:61)")
    if len(rates) != 4 or len(strides) != 4 or rates[2] != rates[3] or strides[2] != strides[3]:
        raise ValueError("rates / strides must be [1, 1, r, r] / [1, 1, s, s] (deform_conv.cc:409-438)")
    cout, cin_g, kh, kw = filter.shape
    if num_groups < 1 or x.shape[-1] != cin_g * num_groups or cout % num_groups != 0:
        raise ValueError("filter [%d, %d, ...] does not fit %d input channels in %d groups (deform_conv.cc:426-438)"
                         % (cout, cin_g, x.shape[-1], num_groups))
    if offset.shape[-1] != 2 * kh * kw * deformable_group:
        raise ValueError("offset must have 2*kh*kw*deformable_group = %d channels (deform_conv.cc:116), got %d"
                         % (2 * kh * kw * deformable_group, offset.shape[-1]))
    if padding == "VALID":
        if int(strides[2]) != 1:
            raise ValueError("padding='VALID' is composed from the SAME kernels for stride 1 only")
        H, W = x.shape[1], x.shape[2]
        ho, wo = H - kh + 1, W - kw + 1
        if tuple(offset.shape[1:3]) != (ho, wo):
            raise ValueError("VALID offsets must be [N, %d, %d, .] (deform_conv.cc:473-479), got %s" % (ho, wo, tuple(offset.shape)))
        pt, pl = (kh - 1) // 2, (kw - 1) // 2
        full = torch.nn.functional.pad(offset, (0, 0, pl, W - wo - pl, pt, H - ho - pt))
        y = deform_conv_op(x, filter, full, rates, "SAME", strides, num_groups, deformable_group, bias=bias, relu=relu)
        return y[:, pt:pt + ho, pl:pl + wo, :].contiguous()
    if padding != "SAME":
        raise ValueError("padding must be 'SAME' or 'VALID'")
    if num_groups != 1:
        G, dg, C = num_groups, deformable_group, x.shape[-1]
        if dg % G != 0 and G % dg != 0:
            raise ValueError("num_groups = %d and deformable_group = %d: one must divide the other" % (G, dg))
        T2 = 2 * kh * kw
        outs = []
        for gi in range(G):
            xs = x[..., gi * cin_g:(gi + 1) * cin_g].contiguous()
            if dg % G == 0:                          # the slice covers dg / G whole deformable groups
                d0, dl = gi * (dg // G), dg // G
            else:                                    # the slice lies inside ONE deformable group
                d0, dl = gi // (G // dg), 1
            offs = offset[..., d0 * T2:(d0 + dl) * T2].contiguous()
            fg = filter[gi * (cout // G):(gi + 1) * (cout // G)]
            bg = None if bias is None else bias[gi * (cout // G):(gi + 1) * (cout // G)]
            outs.append(deform_conv_op(xs, fg, offs, rates, "SAME", strides, 1, dl, bias=bg, relu=relu))
        return torch.cat(outs, dim=-1)
    cin = cin_g
    w1x1 = getattr(filter, "_danhip_hwio", None)       # (deform_conv_2d: the variable's GEMM operand where the store keeps one)
    if w1x1 is None or tuple(w1x1.shape) != (1, 1, kh * kw * cin, cout):
        w1x1 = filter.permute(2, 3, 1, 0).reshape(1, 1, kh * kw * cin, cout).contiguous()      # OIHW -> HWIO over k = tap*C + c
    if cout % 8 == 0:
        # DeformConvOp / DeformConvBackpropOp as single library calls (danhip_deform_conv_{fwd,bwd})
        y = ops.deform_conv(x, w1x1, bias, offset, kh, kw, stride=int(strides[2]), dilation=int(rates[2]), deformable_group=deformable_group,
                            relu=relu)
        if ops.TRACE is not None and relu:
            ops.TRACE[id(filter)] = y.detach()
        return y
    # ragged Cout: the two halves as separate ops (same kernels; the conv wrapper pads the output-gradient channels)
    S = ops.deform_sample(x, offset, kh, kw, stride=int(strides[2]), dilation=int(rates[2]), deformable_group=deformable_group)
    return ops.conv2d(S, w1x1, bias, stride=1, relu=relu)


def deform_conv_2d(inputs, num_outputs, kernel_size_h=3, kernel_size_w=3, stride=1, dilate_rate=1, deformable_group=1,
                   data_format="channels_last", no_bias=True, kernel_initializer="glorot_oihw", name=None, variables=None, relu=False):
    """custom_op.deform_conv_2d (utility/custom_op.py:128-146): zero-initialised offset conv (tf.layers default name
    'conv2d'), OIHW kernel variable 'kernel', optional 'bias'.  `variables` is the VariableStore (tf.get_variable)."""
    from .. import ops
    if data_format != "channels_last":
        raise ValueError("the MI355X build is NHWC-native")
    name = name or "deform_conv"
    cin = inputs.shape[-1]
    noff = 2 * deformable_group * kernel_size_h * kernel_size_w
    ow = variables.get(name + "/conv2d/kernel", (kernel_size_h, kernel_size_w, cin, noff), "zeros")
    ob = variables.get(name + "/conv2d/bias", (noff,), "zeros")
    offset = ops.conv2d(inputs, ow, ob, stride=stride, relu=False)
    if ops.TRACE is not None:                          # tests: the sampling positions of this pass (imposed on the oracle graph)
        ops.TRACE.setdefault("offsets", {})[id(ow)] = offset.detach()
    kernel = variables.get(name + "/kernel", (num_outputs, cin, kernel_size_h, kernel_size_w), kernel_initializer)
    # the GEMM operand of the OIHW variable: a block of the trainer's flat buffers (the variable is then its permuted view), or the copy
    # cached per weight version for gradient-free passes, or None (deform_conv_op transposes per call: plain autograd)
    kernel._danhip_hwio = variables.fuse((name + "/kernel",), axis="hwio")
    bias = None if no_bias else variables.get(name + "/bias", (num_outputs,), "zeros")
    return deform_conv_op(inputs, kernel, offset, [1, 1, dilate_rate, dilate_rate], "SAME", [1, 1, stride, stride], 1, deformable_group, bias=bias, relu=relu)


class _DeformPSROIPool(torch.autograd.Function):
    @staticmethod
    def forward(ctx, data, rois, trans, cfg):
        spatial_scale, output_dim, group_size, pooled_size, part_size, sample_per_part, trans_std, no_trans = cfg
        B, C, H, W = data.shape
        R = rois.shape[0]
        ncls = 1 if no_trans else trans.shape[1] // 2
        top = torch.empty((R, output_dim, pooled_size, pooled_size), dtype=torch.float32, device=data.device)
        cnt = torch.empty_like(top)
        call("danhip_deform_psroi_pool_fwd", ptr(data), ptr(rois), ptr(trans), ptr(top), ptr(cnt), R, C, H, W, output_dim, group_size, pooled_size,
             part_size, sample_per_part, float(spatial_scale), float(trans_std), int(no_trans), ncls, stream())
        ctx.save_for_backward(data, rois, trans, cnt)
        ctx.cfg = cfg
        ctx.mark_non_differentiable(cnt)
        return top, cnt

    @staticmethod
    def backward(ctx, grad, _):
        data, rois, trans, cnt = ctx.saved_tensors
        spatial_scale, output_dim, group_size, pooled_size, part_size, sample_per_part, trans_std, no_trans = ctx.cfg
        B, C, H, W = data.shape
        R = rois.shape[0]
        ncls = 1 if no_trans else trans.shape[1] // 2
        dd = torch.empty_like(data)
        dt = torch.empty_like(trans)
        call("danhip_deform_psroi_pool_bwd", ptr(grad.contiguous()), ptr(cnt), ptr(data), ptr(rois), ptr(trans), ptr(dd), ptr(dt), B, R, C, H, W,
             output_dim, group_size, pooled_size, part_size, sample_per_part, float(spatial_scale), float(trans_std), int(no_trans), ncls, stream())
        return dd, None, (None if no_trans else dt), None


def deform_psroi_pool(data, rois, trans, spatial_scale, output_dim, group_size, pooled_size, part_size=0, sample_per_part=1, trans_std=0.0,
                      no_trans=False):
    """custom_op.deform_psroi_pool (utility/custom_op.py:93; DeformPSROIPool cpp/Deform/deform_psroi_pooling_op.cc:37-79) with the TF op's
    tensors: data fp32 NCHW, rois fp32 [R,5], trans fp32 [R,2*num_classes,part_size,part_size] -> (top_data, mapping_channel); gradients
    w.r.t. data and trans as registered at custom_op.py:97-126 (none for rois)."""
    for t, nd, what in ((data, 4, "data"), (rois, 2, "rois"), (trans, 4, "trans")):
        if t.dim() != nd:
            raise ValueError("%s must be %d-dimensional (deform_psroi_pooling_op.cc:182-192)" % (what, nd))
    if data.dtype != torch.float32 or rois.dtype != torch.float32 or trans.dtype != torch.float32:
        raise ValueError("deform_psroi_pool: float32 tensors (the op's T)")
    cfg = (float(spatial_scale), int(output_dim), int(group_size), int(pooled_size), int(part_size), int(sample_per_part), float(trans_std), bool(no_trans))
    return _DeformPSROIPool.apply(data.contiguous(), rois.contiguous(), trans.contiguous(), cfg)
