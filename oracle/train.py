"""Loss / hard-negative mining / optimizer oracle (PyTorch-CPU fp32) — restates the model_fn bodies of
train_sfd.py, train_pb.py and train_dan.py.  Oracle only (see oracle/__init__.py)."""
import torch
import torch.nn.functional as F


def modified_smooth_l1(pred, target, sigma=1.0):
    """train_sfd.py:224-243 / train_dan.py:226-245 (inside/outside weights = 1)."""
    s2 = sigma * sigma
    d = pred - target
    sign = (d.abs() < 1.0 / s2).to(pred.dtype)
    return (d * d) * (0.5 * s2) * sign + (d.abs() - 0.5 / s2) * (sign - 1.0).abs()


def hard_neg_mask(cls_pred, cls_targets, negative_ratio=3.0, at_least_one=False):
    """Per-image hard-negative mining — train_sfd.py:350-384 (S3FD/PB: k=min(3*pos, neg)),
    train_dan.py:286-324 (DAN: additionally max(k,1)).
    cls_pred [B,A,2] logits, cls_targets [B,A] in {1,0,-1}.  Returns final_mask [B,A] bool, positive_mask.
    k = 0 is undefined in the reference (gather_nd index -1); the build defines it as "select no negatives"."""
    B, A, _ = cls_pred.shape
    pos = cls_targets > 0
    neg = cls_targets == 0
    n_pos = pos.sum(-1)
    n_neg = neg.sum(-1)
    k = torch.minimum((negative_ratio * n_pos.to(torch.float32)).to(torch.int32), n_neg.to(torch.int32))
    if at_least_one:
        k = torch.clamp(k, min=1)
    p_bg = F.softmax(cls_pred.to(torch.float32), dim=-1)[:, :, 0]
    score = torch.where(neg, 0.0 - p_bg, 0.0 - torch.ones_like(p_bg))
    topk, _ = torch.sort(score, dim=-1, descending=True)
    sel = torch.zeros_like(neg)
    for b in range(B):
        if int(k[b]) >= 1:
            thr = topk[b, int(k[b]) - 1]
            sel[b] = score[b] >= thr
    final = (neg & sel) | pos
    return final, pos, score, k


def detection_loss(cls_pred, loc_pred, cls_targets, loc_targets, negative_ratio=3.0, at_least_one=False):
    """CE * (neg_ratio+1) + mean-over-positives smooth-L1 — train_sfd.py:386-417, train_dan.py:470-478.
    cls_pred [B,A,2], loc_pred [B,A,4]; returns (ce, loc, final_mask)."""
    final, pos, _, _ = hard_neg_mask(cls_pred.detach(), cls_targets, negative_ratio, at_least_one)
    logits = cls_pred[final]
    labels = torch.clamp(cls_targets[final], 0, 2).to(torch.int64)
    ce = F.cross_entropy(logits, labels, reduction="mean") * (negative_ratio + 1.0)
    lp = loc_pred[pos]
    lt = loc_targets[pos]
    loc = modified_smooth_l1(lp, lt).sum(-1).mean()
    return ce, loc, final


def l2_regularizer(params, weight_decay=5e-4):
    """train_sfd.py:419-427: wd * sum(l2_loss(var)) over non-bias, non-bn vars; l2_norm_layer weights * 0.2.
    tf.nn.l2_loss(t) = sum(t**2)/2."""
    total = 0.0
    for name, v in params.items():
        if "bn" in name:
            continue
        if "l2_norm_layer" in name:
            total = total + 0.2 * 0.5 * (v * v).sum()
        elif "/bias" not in name:
            total = total + 0.5 * (v * v).sum()
    return weight_decay * total


def momentum_sgd_step(params, grads, momenta, lr, momentum=0.9):
    """tf.train.MomentumOptimizer (train_sfd.py:447) with gradient_multipliers: x2 for '/bias' vars (:436-439).
    v <- m*v + g ; w <- w - lr*v."""
    for name in params:
        g = grads[name]
        if "/bias" in name:
            g = g * 2.0
        momenta[name].mul_(momentum).add_(g)
        params[name].sub_(lr * momenta[name])
