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
        b = self.backbone
        x = sfd_net.prepare_input(images_u8)
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
        return boxes, torch.softmax(cls, dim=-1)[..., 1]
