"""Times the target encoder (IoU -> matching -> encode) for one 640x640 batch: per-image calls vs danhip_encode_anchors_batched."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dan_amd import synthetic
from dan_amd.train_sfd import AnchorConfig

dev = torch.device("cuda:0")
B, S = 16, 640
for faces in (5, 40, 300):
    gts = [g.to(dev) for g in synthetic.make_gt_boxes(B, S, S, seed=3, max_faces=faces)]
    cfg = AnchorConfig(S, S, dev)
    ymin, xmin, ymax, xmax, inside = cfg.anchors

    def per_image():
        return [cfg.enc.encode_anchors(g, ymin, xmin, ymax, xmax, inside, match_mining=True) for g in gts]

    def batched():
        return cfg.enc.encode_anchors_batch(gts, ymin, xmin, ymax, xmax, inside, match_mining=True)

    for name, fn in (("per-image", per_image), ("batched", batched)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            fn()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        print("max_faces %4d  (total gt %5d)  %-10s %8.3f ms / batch of %d   %9.0f images/s" % (faces, sum(g.shape[0] for g in gts), name, ms, B, B / ms * 1e3))
