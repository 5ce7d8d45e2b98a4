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
                           trainging, thres, ignore_thres, seed=0, counter0=0):
    """custom_op.dynamic_anchor_routing (utility/custom_op.py:52).  Tensors may carry a leading batch dimension
    ([B,N,4] / [B,N]); img_height / img_width are accepted and ignored exactly as the reference kernel does.
    Training mode draws u(b,i) from the counter-based stream keyed by (seed, counter0) (the reference's
    std::random_device stream is not reproducible).  Returns (mask_out int32, decode_out fp32) of the input's shape."""
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
             int(feat_strides), B, float(thres), float(ignore_thres), int(seed), int(counter0), ptr(mo), ptr(do), ptr(ws), nws, stream())
    else:
        call("danhip_dynamic_anchor_routing_eval", ptr(a), ptr(g), ptr(l), ptr(m), N, int(feat_height), int(feat_width), int(anchor_depth),
             int(feat_strides), B, ptr(mo), ptr(do), ptr(ws), nws, stream())
    return mo, do
