"""Single-scale S3FD inference step (forward + softmax + decode) alone, for a kernel profile: python tools/bench_eval_step.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dan_amd import synthetic
from dan_amd.train_sfd import AnchorConfig, SFDModel

dev = torch.device("cuda:0")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
model = SFDModel(device=dev, seed=1)
anchors = AnchorConfig(640, 640, dev)
imgs = synthetic.make_images(16, 640, 640, dev, seed=3)
for _ in range(3):
    model.predict(imgs, anchors)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    model.predict(imgs, anchors)
e1.record()
torch.cuda.synchronize()
print("%.3f ms per batch of 16" % (e0.elapsed_time(e1) / iters))
