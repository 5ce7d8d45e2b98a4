"""DAN / DAN-Deform training and inference graphs on libdanhip — the MI355X equivalent of the reference's
train_dan.py / train_dan_deform.py (dan_model_fn :386-532) and eval_dan.py (:299-404)."""
import torch

from . import ops
from .net import danet, sfd_net
from .net.variables import VariableStore
from .train_sfd import ALL_ANCHOR_SCALES, ALL_EXTRA_SCALES, ALL_LAYER_STRIDES, AnchorConfig
from .utility import custom_op

DAN_ANCHOR_RATIOS = [(0.8,)] * 6              # train_dan.py:183


def dan_anchor_config(height, width, device):
    """train_dan.py:181-203: ratio 0.8 anchors, dual-max matching 0.35/0.35 (match_mining=False)."""
    return AnchorConfig(height, width, device, match_threshold=0.35, neg_threshold=0.35, ratios=DAN_ANCHOR_RATIOS)


class DANModel(object):
    def __init__(self, device="cuda", seed=20180817, deform=False):
        self.vs = VariableStore(device=device, seed=seed)
        if deform:
            from .net import danet_deform
            self.backbone = danet_deform.VGG16Backbone("channels_last", variables=self.vs)
        else:
            self.backbone = danet.VGG16Backbone("channels_last", variables=self.vs)

    def forward(self, images_u8):
        """train_dan.py:410-428 -> ((loc1 [B,A,4], cls1 [B,A,2]), (loc2, cls2), feature map sizes)."""
        prec = getattr(self, "precision", "act")
        with sfd_net.precision_scope(prec):
            b = self.backbone
            x = sfd_net.prepare_input(images_u8, prec)
            feats = b.get_featmaps(x, training=True)
            feats = b.build_lfpn(feats, skip_last=3)
            s1 = b.get_features_stage1(feats, name="prediction_modules_stage1")
            s1 = b.build_lfpn(s1, skip_last=3, name="lfpn_stage1")
            n = len(feats)
            stage1 = b.get_predict_module(s1, [1] * n, [1] * n, [1] * n, name="predict_face")
            s2 = b.get_features_stage2(s1, feats, name="prediction_modules_stage2")
            s2 = b.build_lfpn(s2, skip_last=3, name="lfpn_stage2")
            stage2 = b.get_predict_module(s2, [1] * n, [3] + [1] * (n - 1), [1] * n, name="predict_cascade")
            return stage1, stage2, [(f.shape[1], f.shape[2]) for f in feats]

    @torch.no_grad()
    def predict(self, images_u8, anchors, select_thres=0.03):
        """eval_dan.py:344-404: stage-1 boxes of levels 2.. plus the dynamically routed stage-2 boxes of every level.
        Returns (bboxes_pred [B,A',4], cls_pred [B,A']) exactly as fetched at eval_dan.py:99."""
        (loc1, cls1), (loc2, cls2), sizes = self.forward(images_u8)
        enc = anchors.enc
        score1, easy = ops.face_scores(cls1, select_thres)          # eval_dan.py:356, :386
        score2 = ops.face_scores(cls2)                                # eval_dan.py:371
        boxes1 = enc.batch_decode_anchors(loc1, *anchors.anchors[:4])
        if getattr(self, "_scale2", None) is None or self._scale2.device != loc2.device:
            self._scale2 = torch.tensor([20., 20., 10., 10.], dtype=torch.float32, device=loc2.device)   # eval_dan.py:384 (kept: no per-call upload)
        scale = self._scale2
        out_boxes, out_scores = [], []
        off = 0
        per_level = anchors.num_anchors_per_layer
        lvl_boxes, lvl_scores = [], []
        for i, (nl, (fh, fw)) in enumerate(zip(per_level, sizes)):
            sl = slice(off, off + nl)
            mo, do = custom_op.dynamic_anchor_routing(boxes1[:, sl].contiguous(), (loc2[:, sl] / scale).contiguous(), score2[:, sl].contiguous(),
                                                      easy[:, sl].contiguous(),
                                                      fh, fw, anchors.depth[i], ALL_LAYER_STRIDES[i], images_u8.shape[1], images_u8.shape[2], False, 0.03, 0.0)
            lvl_boxes.append(do)
            lvl_scores.append(score2[:, sl] * mo.to(torch.float32))
            off += nl
        first = sum(per_level[:2])
        out_boxes = [boxes1[:, first:]] + lvl_boxes               # eval_dan.py:401-404
        out_scores = [score1[:, first:]] + lvl_scores
        return torch.cat(out_boxes, dim=1), torch.cat(out_scores, dim=1)


from .train_sfd import DetectorTrainer  # noqa: E402

ROUTING_THRES = [0.4, 0.5, 0.6, 0.7, 0.8, 0.9]                 # train_dan.py:440
ROUTING_IGNORE = [0.35, 0.4, 0.45, 0.5, 0.55, 0.6]


def anchor_routing(decoded_bbox, gt_bboxes, gt_labels, easy_mask, feat_sizes, feat_strides, all_num_anchors_depth, num_anchors_per_layer,
                   threshold_per_layer, ignore_threshold_per_layer, image_size, seed, counter0, counter_dev=None):
    """train_dan.py:357-384: split per pyramid level, dynamic_anchor_routing(training) per level (batched over images),
    concat.  -> (final_mask int32 [B,A], final_loc_targets fp32 [B,A,4]); no gradient."""
    masks, targets = [], []
    off = 0
    B = decoded_bbox.shape[0]
    for i, nl in enumerate(num_anchors_per_layer):
        sl = slice(off, off + nl)
        mo, do = custom_op.dynamic_anchor_routing(decoded_bbox[:, sl].contiguous(), gt_bboxes[:, sl].contiguous(), gt_labels[:, sl].contiguous(),
                                                  easy_mask[:, sl].contiguous(), feat_sizes[i][0], feat_sizes[i][1], all_num_anchors_depth[i], feat_strides[i],
                                                  image_size[0], image_size[1], True, threshold_per_layer[i], ignore_threshold_per_layer[i],
                                                  seed=seed, counter0=counter0 + off * B, counter_dev=counter_dev)
        masks.append(mo)
        targets.append(do)
        off += nl
    return torch.cat(masks, dim=1), torch.cat(targets, dim=1)


class DANTrainer(DetectorTrainer):
    """dan_model_fn (train_dan.py:386-532): stage-1 loss against the encoded anchors, dynamic anchor routing of the decoded
    stage-1 boxes into stage-2 targets (x [20,20,10,10], :452), stage-2 loss; mining keeps at least one negative (:302)."""

    def __init__(self, model, anchors, routing_seed=20180817, **kw):
        super().__init__(model, **kw)
        self.anchors = anchors
        self.routing_seed = routing_seed
        dev = model.vs.device
        self._routing_ctr = torch.zeros(1, dtype=torch.int64, device=dev)        # device-resident: advances inside a captured step too
        self._loc_scale2 = torch.tensor([20., 20., 10., 10.], dtype=torch.float32, device=dev)     # train_dan.py:452

    def loss_terms(self, images_u8, loc_targets, cls_targets, matched_gt):
        (loc1, cls1), (loc2, cls2), sizes = self.model.forward(images_u8)
        a = self.anchors
        with torch.no_grad():
            bboxes_pred = a.enc.batch_decode_anchors(loc1.detach(), *a.anchors[:4])                       # :430
            _, easy = ops.face_scores(cls1.detach(), 0.03)                                                # :438-439
            B, A = cls_targets.shape
            final_mask, final_loc = anchor_routing(bboxes_pred, matched_gt, (cls_targets > 0).to(torch.float32), easy, sizes, ALL_LAYER_STRIDES,
                                                   a.depth, a.num_anchors_per_layer, ROUTING_THRES, ROUTING_IGNORE, images_u8.shape[1:3],
                                                   self.routing_seed, 0, counter_dev=self._routing_ctr)
            self._routing_ctr.add_(B * A)
            final_loc = final_loc * self._loc_scale2                                                   # :452
        acc1 = ops.detection_loss(cls1, loc1, cls_targets, loc_targets, ratio=self.negative_ratio, at_least_one=True, scale=self.loss_scale / self.num_towers)
        acc2 = ops.detection_loss(cls2, loc2, final_mask, final_loc, ratio=self.negative_ratio, at_least_one=True, scale=self.loss_scale / self.num_towers)
        self.last_routing = (final_mask, final_loc)
        return [("stage1", 1.0, acc1), ("stage2", 1.0, acc2)]


def encode_batch_dan(anchors, gt_boxes_list):
    """anchor_encoder_fn of train_dan.py:206 per image -> (loc_targets [B,A,4], cls_targets [B,A] int32, matched_gt [B,A,4])."""
    ymin, xmin, ymax, xmax, inside = anchors.anchors
    t, l, _, m = anchors.enc.encode_anchors_batch(gt_boxes_list, ymin, xmin, ymax, xmax, inside, match_mining=False)
    return t, l, m
