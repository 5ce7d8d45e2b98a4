"""S3FD training / inference graph on libdanhip — the MI355X equivalent of the reference's train_sfd.py
(input_pipeline anchor configuration :171-222 and sfd_model_fn :261-467) and eval_sfd.py's inference graph (:232-283).
"""
import os

import torch

from . import ops
from .net import sfd_net
from .net.variables import VariableStore
from .trainer import FlatParams, GradBuckets, lr_schedule
from .utility import anchor_manipulator

ALL_ANCHOR_SCALES = [(16.,), (32.,), (64.,), (128.,), (256.,), (512.,)]
ALL_EXTRA_SCALES = [(), (), (), (), (), ()]
ALL_ANCHOR_RATIOS = [(1.,), (1.,), (1.,), (1.,), (1.,), (1.,)]
ALL_LAYER_STRIDES = [4, 8, 16, 32, 64, 128]


def layer_shapes(height, width):
    """Feature pyramid sizes: stride-4 map after two SAME pools, then SAME pool / stride-2 convs (ceil division)."""
    shapes = []
    h, w = -(-height // 4), -(-width // 4)
    for _ in range(6):
        shapes.append((h, w))
        h, w = -(-h // 2), -(-w // 2)
    return shapes


class AnchorConfig(object):
    """train_sfd.py:173-203: encoder thresholds 0.4/0.4 (flags :88-90), prior scaling (0.1,0.1,0.2,0.2), border = image size."""

    def __init__(self, height, width, device, match_threshold=0.4, neg_threshold=0.4, ratios=ALL_ANCHOR_RATIOS):
        self.enc = anchor_manipulator.AnchorEncoder(match_threshold, neg_threshold, [0.1, 0.1, 0.2, 0.2], device=device)
        self.shapes = layer_shapes(height, width)
        hs, ws, ds = [], [], []
        for i in range(6):
            h, w, d = self.enc.get_anchors_width_height(ALL_ANCHOR_SCALES[i], ALL_EXTRA_SCALES[i], ratios[i])
            hs.append(h); ws.append(w); ds.append(d)
        self.depth = ds
        self.anchors = self.enc.get_all_anchors((height, width), hs, ws, ds, [0.5] * 6, self.shapes, ALL_LAYER_STRIDES,
                                                [float(height)] * 6, [False] * 6)
        self.num_anchors_per_layer = [s[0] * s[1] * d for s, d in zip(self.shapes, ds)]
        self.num_anchors = sum(self.num_anchors_per_layer)

    def encode_batch(self, gt_boxes_list, match_mining=True):
        """anchor_encoder_fn of train_sfd.py:206 applied per image -> loc_targets [B,A,4], cls_targets [B,A] int32, match_scores [B,A]."""
        ymin, xmin, ymax, xmax, inside = self.anchors
        t, l, s, _ = self.enc.encode_anchors_batch(gt_boxes_list, ymin, xmin, ymax, xmax, inside, match_mining=match_mining)
        return t, l, s


class SFDModel(object):
    def __init__(self, device="cuda", seed=20180817):
        self.vs = VariableStore(device=device, seed=seed)
        self.backbone = sfd_net.VGG16Backbone("channels_last", variables=self.vs)

    def forward(self, images_u8):
        """train_sfd.py:286-304: returns (location_pred [B,A,4], cls_pred [B,A,2]) fp32."""
        prec = getattr(self, "precision", "act")
        self.backbone.fp32_heads = prec == "mixed"          # 16-bit backbone, fp32 L2-norm taps + heads (inference only)
        with sfd_net.precision_scope(prec):
            x = sfd_net.prepare_input(images_u8, prec if prec in ("fp32", "split") else "act")
            feats = self.backbone.get_featmaps(x, training=True)
            return self.backbone.multibox_head(feats, [1] * 6, [3] + [1] * 5, [1] * 6)

    @torch.no_grad()
    def predict(self, images_u8, anchors):
        """eval_sfd.py:262-283: (bboxes_pred [B,A,4], face scores [B,A])."""
        loc, cls = self.forward(images_u8)
        boxes = anchors.enc.batch_decode_anchors(loc, *anchors.anchors[:4])
        return boxes, ops.face_scores(cls)


def _tree_clone(t):
    if torch.is_tensor(t):
        return t.clone()
    if isinstance(t, dict):
        return {k: _tree_clone(v) for k, v in t.items()}
    if isinstance(t, (list, tuple)):
        return type(t)(_tree_clone(v) for v in t)
    return t


def _tree_copy(dst, src):
    if torch.is_tensor(dst):
        if dst.data_ptr() != src.data_ptr():
            dst.copy_(src)
    elif isinstance(dst, dict):
        for k in dst:
            _tree_copy(dst[k], src[k])
    elif isinstance(dst, (list, tuple)):
        for d, s_ in zip(dst, src):
            _tree_copy(d, s_)


class DetectorTrainer(object):
    """Shared body of the reference's *_model_fn training branch (train_sfd.py:261-467, train_pb.py:350-520,
    train_dan.py:386-532): forward -> loss terms (hard-negative mining + CE*(ratio+1) + smooth-L1, each with its weight)
    -> backward into the flat gradient buffer (+ bucketed all-reduce) -> fused momentum SGD (L2 term, bias gradient x2).
    `world` ranks each see batch/world images; every term's gradient carries the 1/world of tf_replicate_model_fn.py:297-302.
    Subclasses implement loss_terms(images_u8, *targets) -> list of (name, acc4 tensor)."""

    def __init__(self, model, world=1, weight_decay=5e-4, negative_ratio=3.0, momentum=0.9, base_lr=1e-3, init_hw=(64, 64),
                 lr_boundaries=(1000, 80000, 100000), lr_factors=(0.1, 1.0, 0.1, 0.01), loss_scale=None, dynamic_loss_scale=None,
                 loss_scale_growth_interval=1000, ops_ctx=None):
        self.model = model
        # the ops' steering state this trainer runs under (kernel-form switches, diagnostic sinks, the per-step hooks armed below): the
        # context active at construction unless one is handed in - every step runs inside it (ops.OpsContext)
        self.ops_ctx = ops_ctx if ops_ctx is not None else ops.context()
        # a context handed in explicitly is entered by every step; otherwise a step runs in whatever context is active at CALL time, so
        # `with ops.use_context(...): tr.train_step(...)` steers it as the OpsContext docstring promises
        self._own_ctx = ops_ctx is not None
        # fp16 build: activation gradients below 6e-8 flush to zero, so the backward pass runs on loss * loss_scale (the factor enters
        # through the loss terms' `scale`: their backward ignores the upstream seed) and the
        # fused optimizer divides the (fp32) weight gradients again; bf16 has fp32's exponent range and needs none
        from ._lib import ACT_NAME
        self.loss_scale = float(loss_scale) if loss_scale is not None else (1024.0 if ACT_NAME == "fp16" else 1.0)
        # dynamic_loss_scale: the scale lives on the device ({scale, clean steps, growth interval, flag}); the loss terms multiply their
        # gradients by it, the optimizer entry checks the gradients for inf / NaN, skips such a step and applies GradScaler's rule
        # (x0.5 after a skipped step, x2 after `interval` clean ones).  `loss_scale` is then the initial value.
        self.ls_state = None
        if dynamic_loss_scale is None:                        # default: on for the fp16 build unless a static scale was asked for
            dynamic_loss_scale = ACT_NAME == "fp16" and loss_scale is None
        if dynamic_loss_scale:
            self.ls_state = torch.tensor([self.loss_scale, 0.0, float(loss_scale_growth_interval), 0.0], dtype=torch.float32, device=model.vs.device)
            self.loss_scale = 1.0                             # host-side factor of the loss terms; the device scalar carries the scale
        self.world = world
        self.towers = 1                                       # towers of THIS rank inside one step (train_step_towers)
        self.negative_ratio = negative_ratio
        self.momentum = momentum
        self.base_lr = base_lr
        self.lr_boundaries, self.lr_factors = lr_boundaries, lr_factors
        dev = model.vs.device
        with torch.no_grad():       # create every variable (shapes do not depend on the image size)
            model.forward(torch.zeros((1, init_hw[0], init_hw[1], 3), dtype=torch.uint8, device=dev))
        self.flat = FlatParams(model.vs, weight_decay)
        self.buckets = GradBuckets(self.flat)
        # Round 5, OPT-IN (DANHIP_OPT_OVERLAP=1): the optimizer runs BUCKET BY BUCKET on the buckets' stream, right behind each bucket's
        # all-reduce (or, on one GPU, as soon as both backward streams have produced the bucket's gradients), together with the re-packing of
        # that bucket's convolution weights - instead of one update + one re-packing pass over all parameters after backward.  What stays
        # behind the last backward kernel is then the small tail bucket (GradBuckets: conv1_1 .. conv3_1).  Needs stream-ordered collectives
        # (RCCL or none) and a static loss scale (the dynamic one inspects ALL gradients before any update).  MEASURED ON ONE GPU it LOSES:
        # same box, alternating, S3FD batch 16: 12.68-12.74 ms with it against 12.60-12.68 without (profiles/r5/ab_vs_round4_same_box.txt) -
        # the persistent convolution kernels leave no CU for the update to hide on, so it only adds launches and HBM traffic mid-backward.
        # With more than one rank the end of the step is exposed to the last all-reduce anyway and the form may pay; nobody has measured it
        # (one-GPU boxes), so the default stays the single pass.
        self.opt_overlap = (os.environ.get("DANHIP_OPT_OVERLAP", "0") == "1" and self.ls_state is None and self.flat.g.is_cuda
                            and (not self.buckets.enabled or self.buckets.device_collectives))
        if self.opt_overlap:
            self.buckets.enable_local()
            self.buckets.on_bucket = self._bucket_opt
            self._bucket_plans = [self.flat.range_plan(s, e) for s, e in self.buckets.bounds]
            w0 = self.flat.w.data_ptr()
            self._bucket_select = [(lambda p, lo=w0 + 4 * s, hi=w0 + 4 * e: lo <= p.data_ptr() < hi) for s, e in self.buckets.bounds]
        self.param_name = {id(p): n for n, p in model.vs.named()}
        self.param_name.update({id(t): key[0] for key, t in model.vs.fused.items()})
        self.step_no = 0
        self.last = None
        self._graph = None

    @property
    def num_towers(self):
        """number_of_towers of tf_replicate_model_fn.py:615-631 (_scale_loss): every tower's loss carries 1 / (ranks x towers per rank)."""
        return self.world * self.towers

    def _hook(self, p):
        n = self.param_name.get(id(p))
        if n is not None:
            self.buckets.ready(n)

    def _bucket_opt(self, b, s, e):
        """GradBuckets.on_bucket: momentum-SGD of the flat range [s, e) and the 16-bit re-packing of its convolution weights, on the
        buckets' stream (the kernels that needed the old packings of these layers were launched before the events this stream waited for)."""
        self.flat.sgd_range(s, e, self._bucket_plans[b], self._step_lr, self.momentum, grad_scale=1.0 / self.loss_scale)
        if not self._epoch_bumped:
            ops.WEIGHT_EPOCH += 1                             # (the packings refreshed below are stamped with the new epoch)
            self._epoch_bumped = True
        ops.repack_all(key=("bucket", id(self), b), select=self._bucket_select[b])

    def loss_terms(self, images_u8, *targets):
        raise NotImplementedError

    def train_step(self, images_u8, *targets):
        """One optimisation step on this rank's shard.  Returns the list of (name, weight, device 4-vector
        [ce_sum, n_selected, loc_sum, n_pos]) loss terms (no host sync).  After enable_graph() the step is one hipGraph launch."""
        if self._graph is None and getattr(self, "_recapture", False):      # restore() dropped the captured step: capture again (same static buffers)
            self._recapture = False
            self._capture()
        if self._graph is not None:
            return self._graph_step(images_u8, *targets)
        return self._eager_step(images_u8, *targets)

    def train_step_towers(self, towers):
        """One optimisation step over SEVERAL towers on this rank, the way the reference places more towers than it has devices
        (tf_replicate_model_fn.py:504-560 _get_loss_towers loops over the shards; :297-343 sums their gradients): `towers` is a list of
        train_step argument tuples (contiguous shards of this rank's part of the global batch); every tower differentiates
        loss_tower / (ranks x len(towers)) into the SAME flat gradient buffer (the backward kernels accumulate), the bucketed all-reduce
        follows the LAST tower's backward, then one optimizer step.  Eager launches only.  Used by bench.py's strong-scaling series
        (a fixed global batch on fewer GPUs than it has 16-image shards)."""
        if self._graph is not None:
            raise RuntimeError("train_step_towers runs eager steps: build the trainer without enable_graph()")
        self.towers = len(towers)
        try:
            return self._eager_towers(towers)
        finally:
            self.towers = 1

    def _eager_step(self, images_u8, *targets):
        return self._eager_towers([(images_u8,) + tuple(targets)])

    def _eager_towers(self, towers):
        if self._own_ctx:
            with ops.use_context(self.ops_ctx):
                return self._eager_towers_in_context(towers, self.ops_ctx)
        return self._eager_towers_in_context(towers, ops.context())

    def _eager_towers_in_context(self, towers, octx):
        self.flat.zero_grad()
        self.buckets.begin_step()
        self._step_lr = lr_schedule(self.step_no, self.base_lr, self.lr_boundaries, self.lr_factors)
        self._epoch_bumped = False
        if self.opt_overlap:
            self.flat.l2.zero_()                              # (every bucket's update adds its share of the L2 term)
        tower_terms = []
        for t, args in enumerate(towers):
            armed = False
            try:                                              # (the hooks are disarmed whichever of forward / backward raises: ADVICE r5)
                # gradients become final in the last tower's backward: only then may a bucket leave
                octx.GRAD_READY_HOOK = self._hook if (self.buckets.active and t == len(towers) - 1) else None
                octx.LOSS_SCALE_DEV = self.ls_state[0:1] if self.ls_state is not None else None
                terms = self.loss_terms(*args)
                tower_terms.append(terms)
                accs = [a[2] for a in terms]
                # weight gradients on a second stream, next to the data gradients — also beside the bucketed all-reduce's stream when the
                # collectives are device-side (RCCL; checked on hardware with a one-rank communicator, tests/test_zz_ddp_gpu.py).  gloo's
                # host-staged all-reduce stalled in that combination (tools/debug_dp_overlap.py), so gloo groups keep one compute stream
                if not self.buckets.enabled or self.buckets.device_collectives or os.environ.get("DANHIP_WGRAD_STREAM_DP") == "1":
                    ops.wgrad_overlap_begin()
                    armed = True
                torch.autograd.backward(accs, [torch.ones_like(a) for a in accs])
            except BaseException:
                if armed:
                    ops.wgrad_overlap_join()                  # never leave the second stream armed behind a failed step
                raise
            finally:
                octx.GRAD_READY_HOOK = None
                octx.LOSS_SCALE_DEV = None
            if t < len(towers) - 1:
                ops.wgrad_overlap_join()                      # (the next tower's forward reuses this one's activation buffers)
        self.buckets.finish()
        ops.wgrad_overlap_join()
        # the L2 term is rank-independent: its gradient wd*w is added inside the fused optimizer kernel
        if not self.opt_overlap:                              # (else every bucket has been updated and re-packed behind its reduction)
            self.flat.sgd_step(self._step_lr, self.momentum, grad_scale=1.0 / self.loss_scale, dynamic_state=self.ls_state)
        self.step_no += 1
        self.last = terms
        self.last_towers = tower_terms                        # every tower's terms: loss_values() reports their mean
        return terms

    # ---- hipGraph capture of the whole step (forward, backward, bucketed all-reduce, optimizer, weight repack): one launch per step
    # instead of ~600 (S3FD) ... ~3000 (DAN) kernel launches from Python.  The data-parallel step is captured too when the collectives
    # are device-side (RCCL): the buckets' side stream forks from the capturing stream through the gradient-ready events and joins it
    # again in GradBuckets.finish(), so every rank replays [backward kernels || all-reduce of finished buckets] -> optimizer as one graph;
    # the host-side bucket bookkeeping runs once, at capture.  gloo (host-staged) collectives cannot be captured.
    # The learning rate is a kernel argument, so the graph is re-captured when the schedule moves.
    def enable_graph(self, images_u8, *targets, warmup=2):
        if self.buckets.enabled and not self.buckets.device_collectives:
            raise RuntimeError("graph capture of the data-parallel step needs device-side collectives (DANHIP_DP_TRANSPORT=rccl), not gloo")
        if self.buckets.enabled and self.buckets.check:
            raise RuntimeError("DANHIP_DP_CHECK compares on the host: not capturable")
        self._graph = None
        self._static = _tree_clone((images_u8,) + tuple(targets))
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                        # warm-up on a side stream (allocator / lazy init), real steps
            for _ in range(warmup):
                self._eager_step(*self._static)
        torch.cuda.current_stream().wait_stream(side)
        self._capture()

    def _capture(self):
        self._graph_lr = lr_schedule(self.step_no, self.base_lr, self.lr_boundaries, self.lr_factors)
        g = torch.cuda.CUDAGraph()
        step_no = self.step_no
        # The collectives are the library's own RCCL calls on the buckets' stream (trainer.RcclComm): recorded like kernels, no framework
        # thread polls events meanwhile.  RCCL itself may keep helper threads that touch the HIP runtime (proxy / progress), so the
        # capture-safety check is restricted to this thread when a communicator exists.
        mode = "thread_local" if self.buckets.enabled else "global"
        with torch.cuda.graph(g, capture_error_mode=mode):
            terms = self._eager_step(*self._static)
        self.step_no = step_no                               # recording does not execute: nothing was stepped
        self._graph, self._graph_terms = g, terms

    def _graph_step(self, images_u8, *targets):
        if lr_schedule(self.step_no, self.base_lr, self.lr_boundaries, self.lr_factors) != self._graph_lr:
            self._capture()
        _tree_copy(self._static, (images_u8,) + tuple(targets))
        self._graph.replay()
        if self.buckets.watch is not None:                   # failure detection of the exchange inside the replayed step (trainer.CommWatch)
            self.buckets.watch.tick(torch.cuda.current_stream())
        self.step_no += 1
        self.last = self._graph_terms
        return self._graph_terms

    def close(self):
        """Drops the captured step (its nodes reference the communicator and the static buffers) and waits for the device: call before
        the RCCL communicator or the process group is destroyed (trainer.shutdown_distributed does)."""
        self._graph = None
        self._graph_terms = None
        self._recapture = False
        if self.flat.w.is_cuda:
            torch.cuda.synchronize()

    # ---- checkpoint / resume (the Estimator's save_checkpoints_steps + resume-from-model_dir behaviour, train_dan.py:549-556)
    def save(self, prefix, model_scope, checksums=False):
        """Variables, batch-norm moving statistics, Momentum slots and global_step under the reference's names."""
        from .utility import checkpoint
        torch.cuda.synchronize() if self.flat.w.is_cuda else None
        checkpoint.save_checkpoint(self.model.vs, prefix, model_scope, checksums=checksums, trainer=self)

    def restore(self, checkpoint_path, model_scope, strict=True):
        from .utility import checkpoint
        return checkpoint.restore_checkpoint(self.model.vs, checkpoint_path, model_scope, trainer=self, strict=strict)

    def loss_values(self):
        """{name: (cross_entropy, loc_loss)} per term + 'l2' + 'total' as python floats (synchronises)."""
        out, total = {}, 0.0
        towers = getattr(self, "last_towers", None)
        if not towers or towers[-1] is not self.last:         # (a replayed graph step, or a caller that assigned .last itself)
            towers = [self.last]
        # several towers on this rank (train_step_towers): the reference reports the mean of the tower losses, each normalised by its own
        # counts (tf_replicate_model_fn.py:661-663) — not the last tower's
        for k, (name, weight, _) in enumerate(towers[0]):
            ce = loc = 0.0
            for terms in towers:
                ce_sum, n_sel, loc_sum, n_pos = [float(v) for v in terms[k][2].tolist()]
                ce += (self.negative_ratio + 1.0) * ce_sum / max(n_sel, 1.0) / len(towers)
                loc += loc_sum / max(n_pos, 1.0) / len(towers)
            out[name] = (ce, loc)
            total += weight * (ce + loc)
        out["l2"] = float(self.flat.l2.item())
        out["total"] = total + out["l2"]
        return out


class SFDTrainer(DetectorTrainer):
    """sfd_model_fn (train_sfd.py:261-467) + Momentum optimizer."""

    def loss_terms(self, images_u8, loc_targets, cls_targets):
        loc, cls = self.model.forward(images_u8)
        acc = ops.detection_loss(cls, loc, cls_targets, loc_targets, ratio=self.negative_ratio, at_least_one=False, scale=self.loss_scale / self.num_towers)
        return [("face", 1.0, acc)]

    def losses(self):
        """(cross_entropy, loc_loss, l2_loss, total) as python floats (synchronises)."""
        v = self.loss_values()
        return v["face"][0], v["face"][1], v["l2"], v["total"]
