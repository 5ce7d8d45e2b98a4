#!/usr/bin/env python3
"""The fp32 evaluation graph (S3FD 640 x 640, batch 4: model.precision = "fp32") a few times - run under rocprofv3 --kernel-trace --stats to see
where its time goes (tools/prof_db.py <db> <iterations>).  usage: python tools/prof_f32_eval.py [iters=5]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from dan_amd import synthetic
from dan_amd.train_sfd import AnchorConfig, SFDModel

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda:0")
model = SFDModel(device=dev)
anchors = AnchorConfig(640, 640, dev)
imgs = synthetic.make_images(4, 640, 640, dev, seed=1)
model.precision = "fp32"
model.predict(imgs, anchors)
torch.cuda.synchronize()
import time
t0 = time.perf_counter()
for _ in range(iters):
    model.predict(imgs, anchors)
torch.cuda.synchronize()
print("fp32 eval: %.2f ms per batch of 4 = %.1f img/s" % ((time.perf_counter() - t0) / iters * 1e3, 4 * iters / (time.perf_counter() - t0)))
