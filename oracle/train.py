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


# ------------------------------------------------------------------ data-parallel semantics (tf_replicate_model_fn.py)
def split_batch(tensors, number_of_shards):
    """_split_batch — tf_replicate_model_fn.py:458-498: every tensor is split on dim 0 into `number_of_shards` equal CONTIGUOUS
    pieces (array_ops.split); a batch that does not divide raises (ensure_divisible_by_shards :461-466).
    tensors: one tensor, or a dict / list / tuple of tensors.  Returns a list of `number_of_shards` objects of the same structure."""
    def split_one(t):
        if t.shape[0] % number_of_shards != 0:
            raise ValueError("Batch size {} needs to be divisible by the number of GPUs, which is {}.".format(t.shape[0], number_of_shards))
        return list(torch.chunk(t, number_of_shards, dim=0))

    if torch.is_tensor(tensors):
        return split_one(tensors)
    if isinstance(tensors, dict):
        shards = [{} for _ in range(number_of_shards)]
        for name, t in tensors.items():
            for i, piece in enumerate(split_one(t)):
                shards[i][name] = piece
        return shards
    cols = [split_one(t) for t in tensors]
    return [type(tensors)(c[i] for c in cols) for i in range(number_of_shards)]


def scale_loss(loss, number_of_towers):
    """_scale_loss — tf_replicate_model_fn.py:615-625 with loss_reduction = MEAN (train_dan.py:558): one tower -> unchanged, else
    loss / number_of_towers."""
    return loss if number_of_towers == 1 else loss / (1.0 * number_of_towers)


def dp_step(tower_fn, shards):
    """One data-parallel step's gradient aggregation as replicate_model_fn builds it:
      * every tower i runs the model_fn on its shard and `TowerOptimizer.compute_gradients` differentiates
        `_scale_loss(loss_i)` = loss_i / N (tf_replicate_model_fn.py:297-302) — loss_i is that tower's TOTAL loss, its own
        L2 term and its own per-shard normalisers (positives / mined negatives of the shard) included;
      * `_apply_gathered_gradients` sums the towers' gradients per variable with add_n (:328-343, `_compute_sum_on_device` :633-645)
        and applies them once;
      * the reported loss is the add_n of the scaled tower losses (`_train_spec`, :661-663).
    tower_fn(shard, loss_scale) -> (loss_i [python float or 0-d tensor, UNSCALED], {var name: gradient of loss_i * loss_scale}).
    Returns ({var name: aggregated gradient}, reported loss)."""
    n = len(shards)
    grad_lists, reported = {}, 0.0
    for shard in shards:
        loss_i, grads_i = tower_fn(shard, 1.0 if n == 1 else 1.0 / (1.0 * n))
        reported = reported + scale_loss(loss_i, n)
        for var, g in grads_i.items():
            if g is not None:
                grad_lists.setdefault(var, []).append(g)
    aggregated = {}
    for var, gs in grad_lists.items():                 # math_ops.add_n(values): summed in tower order
        s = gs[0].clone()
        for g in gs[1:]:
            s = s + g
        aggregated[var] = s
    return aggregated, reported
