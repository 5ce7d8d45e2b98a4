"""Single-scale inference graph (forward + softmax + box decode) at batch B, 640 x 640, for one precision (act | split | fp32) and one model
(sfd | pb | dan | dan_deform): images/s, and — run under `rocprofv3 --kernel-trace --stats` — the per-kernel picture of that path.

    python tools/bench_eval_precision.py [--precision split] [--model sfd] [--batch 16] [--iters 10]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dan_amd import synthetic

ap = argparse.ArgumentParser()
ap.add_argument("--precision", default="split")
ap.add_argument("--model", default="sfd")
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--size", type=int, default=640)
ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
dev = torch.device("cuda:0")
if a.model == "sfd":
    from dan_amd.train_sfd import AnchorConfig, SFDModel
    model, anchors = SFDModel(device=dev), AnchorConfig(a.size, a.size, dev)
elif a.model == "pb":
    from dan_amd.train_pb import PBAnchorTargets, PBModel
    model, anchors = PBModel(device=dev), PBAnchorTargets(a.size, a.size, dev).face
else:
    from dan_amd.train_dan import DANModel, dan_anchor_config
    model, anchors = DANModel(device=dev, deform=a.model == "dan_deform"), dan_anchor_config(a.size, a.size, dev)
model.precision = a.precision
imgs = synthetic.make_images(a.batch, a.size, a.size, dev, seed=1)
for _ in range(2):
    model.predict(imgs, anchors)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(a.iters):
    model.predict(imgs, anchors)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / a.iters
print("%s %s batch %d %dx%d: %.3f ms / batch  %.1f images/s" % (a.model, a.precision, a.batch, a.size, a.size, dt * 1e3, a.batch / dt))
