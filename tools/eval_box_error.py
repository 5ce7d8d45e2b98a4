"""how far are the 16-bit inference paths' DECODED BOXES from an fp32 reference?  S3FD, one 640 x 640 synthetic image,
identical (oracle-initialised) weights: the CPU oracle's fp32 boxes against
    act    the library build's 16-bit path (bf16, or fp16 with DANHIP_DTYPE=fp16),
    mixed  16-bit backbone, fp32 L2-norm taps + fp32 head convolutions,
    fp32   the fp32 inference kernels end to end (the path tests/test_eval_f32_gpu.py holds to 1e-4),
as max / p99 / median |delta| in pixels over all 34 125 anchors x 4 coordinates, the share of coordinates within the north-star bound
1e-4 * max(1, |ref|), and each mode's throughput at batch 16.  usage: python tools/eval_box_error.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from dan_amd import _lib, synthetic
from dan_amd.train_sfd import AnchorConfig, SFDModel
from oracle import anchors as OA
from oracle import nets as ON

dev = torch.device("cuda:0")
S = 640
imgs = synthetic.make_images(1, S, S, "cpu", seed=640)
x = ON.preprocess_synthetic(imgs)
P = ON.Params(create=True, seed=21)
with torch.no_grad():
    ON.sfd_forward(P, x)
    g = torch.Generator().manual_seed(99)
    for n in P.t:
        if n.endswith("/bias"):
            P.t[n] = 0.05 * torch.randn(P.t[n].shape, generator=g)
    loc_r, cls_r = ON.sfd_forward(ON.Params(P.t), x)
model = SFDModel(device=dev)
model.vs.load_tf_named(P.t)
anchors = AnchorConfig(S, S, dev)
a4 = [t.cpu().numpy() for t in anchors.anchors[:4]]
ref = OA.decode_anchors(loc_r[0].numpy(), a4, [0.1, 0.1, 0.2, 0.2]).astype(np.float64)
sref = torch.softmax(cls_r[0], dim=-1)[:, 1].numpy()
big = synthetic.make_images(16, S, S, dev, seed=1)
print("S3FD %dx%d, build %s; reference = CPU oracle fp32; %d anchors" % (S, S, _lib.ACT_NAME, ref.shape[0]))
for mode in ("act", "mixed", "fp32"):
    model.precision = mode
    with torch.no_grad():
        loc, cls = model.forward(imgs.to(dev))
    got = OA.decode_anchors(loc[0].float().cpu().numpy(), a4, [0.1, 0.1, 0.2, 0.2]).astype(np.float64)
    d = np.abs(got - ref)
    ok = (d <= 1e-4 * np.maximum(1.0, np.abs(ref))).mean()
    sc = torch.softmax(cls[0].float().cpu(), dim=-1)[:, 1].numpy()
    b = 16 if mode != "fp32" else 4
    for _ in range(2):
        model.predict(big[:b], anchors)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        model.predict(big[:b], anchors)
    torch.cuda.synchronize()
    ips = b * n / (time.perf_counter() - t0)
    print("%-6s boxes: max %.4g px  p99 %.4g px  median %.4g px   within 1e-4*max(1,|ref|): %.2f %%   scores: max |d| %.3g   %.0f img/s (batch %d)"
          % (mode, d.max(), np.percentile(d, 99), np.median(d), 100 * ok, np.abs(sc - sref).max(), ips, b))
model.precision = "act"
