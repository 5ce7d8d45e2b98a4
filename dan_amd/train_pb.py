"""PyramidBox training / inference graph on libdanhip — the MI355X equivalent of the reference's train_pb.py
(pb_model_fn :350-520: forward :396-420, three-head loss :422-470) and eval_pb.py's inference graph."""
import torch

from . import ops
from .net import pb_net, sfd_net
from .net.variables import VariableStore
from .train_sfd import AnchorConfig  # noqa: F401  (same anchor layout: train_pb.py:173-203)


class PBModel(object):
    def __init__(self, device="cuda", seed=20180817):
        self.vs = VariableStore(device=device, seed=seed)
        self.backbone = pb_net.VGG16Backbone("channels_last", variables=self.vs)

    def forward(self, images_u8):
        """train_pb.py:396-420: {'face','head','body'} -> (location_pred [B,A_k,4], cls_pred [B,A_k,2]); the head / body
        heads predict on levels 1.. / 2.. (A_head = 8 525, A_body = 2 125 at 640x640)."""
        prec = getattr(self, "precision", "act")
        with sfd_net.precision_scope(prec):
            b = self.backbone
            x = sfd_net.prepare_input(images_u8, prec)
            feats = b.get_featmaps(x, training=True)
            feats = b.build_lfpn(feats, skip_last=3)
            feats = b.context_pred_module(feats)
            n = len(feats)
            face = b.get_predict_module(feats, [1] + [3] * (n - 1), [3] + [1] * (n - 1), [1] * n, name="predict_face")
            head = b.get_predict_module(feats[1:], [1] * (n - 1), [1] * (n - 1), [1] * (n - 1), name="predict_head")
            body = b.get_predict_module(feats[2:], [1] * (n - 2), [1] * (n - 2), [1] * (n - 2), name="predict_body")
            return {"face": face, "head": head, "body": body}

    @torch.no_grad()
    def predict(self, images_u8, anchors):
        """eval_pb.py inference graph: decoded face boxes + face scores."""
        loc, cls = self.forward(images_u8)["face"]
        boxes = anchors.enc.batch_decode_anchors(loc, *anchors.anchors[:4])
        return boxes, ops.face_scores(cls)


class PBAnchorTargets(object):
    """Anchor targets of the PyramidBox input pipeline (train_pb.py:173-247): face anchors on all 6 levels (small-mining
    match 0.4/0.4), head anchors on levels 1.. matched against the face boxes with anchors shrunk by 2 (dual-max 0.35),
    body anchors on levels 2.. shrunk by 4."""

    def __init__(self, height, width, device):
        self.face = AnchorConfig(height, width, device)
        self.device = device
        n = self.face.num_anchors_per_layer
        self.head_off, self.body_off = n[0], n[0] + n[1]

    def _sub(self, off):
        return tuple(t[off:].contiguous() for t in self.face.anchors)

    def encode_batch(self, gt_boxes_list):
        enc = self.face.enc
        fa, ha, ba = self.face.anchors, self._sub(self.head_off), self._sub(self.body_off)
        gts = [b.to(self.device) for b in gt_boxes_list]
        return {"face": enc.encode_anchors_batch(gts, *fa, match_mining=True)[:3],
                "head": enc.encode_pa_anchors_batch(gts, *ha, 0.35, 0.35, match_mining=False, scale=2.)[:3],       # train_pb.py:218
                "body": enc.encode_pa_anchors_batch(gts, *ba, 0.35, 0.35, match_mining=False, scale=4.)[:3]}       # train_pb.py:223


from .train_sfd import DetectorTrainer  # noqa: E402


class PBTrainer(DetectorTrainer):
    """pb_model_fn (train_pb.py:350-520): total = face + 0.66 * head + 0.33 * body (+ L2), each term
    CE*(ratio+1) over mined rows + smooth-L1 over positives (:440-504)."""
    WEIGHTS = {"face": 1.0, "head": 0.66, "body": 0.33}

    def loss_terms(self, images_u8, targets):
        out = self.model.forward(images_u8)
        terms = []
        for k in ("face", "head", "body"):
            loc, cls = out[k]
            loc_t, cls_t, _ = targets[k]
            acc = ops.detection_loss(cls, loc, cls_t, loc_t, ratio=self.negative_ratio, at_least_one=False, scale=self.loss_scale * self.WEIGHTS[k] / self.num_towers)
            terms.append((k, self.WEIGHTS[k], acc))
        return terms
