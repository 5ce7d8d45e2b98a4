"""Times the whole test-time pipeline of eval_sfd.py / eval_dan.py (origin + flip + multi-scale [+ pyramid] passes, box voting) on
one synthetic 1024x768 image, random-init weights (face-less noise: the voting stage sees only low-score boxes)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dan_amd import eval_dan, eval_sfd, synthetic
from dan_amd.train_sfd import AnchorConfig, SFDModel

dev = torch.device("cuda:0")
img = synthetic.make_images(1, 768, 1024, dev, seed=5)[0]
model = SFDModel(device=dev)
net = eval_dan.Detector(model, lambda h, w, d: AnchorConfig(h, w, d))
for name, fn in (("eval_sfd.detect_image (origin + flip + multi-scale + vote)", lambda: eval_sfd.detect_image(net, img)),
                 ("single-scale detect_face", lambda: eval_dan.detect_face(net, img, 1.0))):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 5
    for _ in range(n):
        out = fn()
    torch.cuda.synchronize()
    print("%-60s %8.1f ms / image   (%d boxes out)" % (name, (time.perf_counter() - t0) / n * 1e3, len(out)))
# the batched pipeline (eval_dan.detect_images: every scale of B same-size images as one forward pass, no host round trip until the end)
for B in (1, 4, 8):
    imgs = synthetic.make_images(B, 768, 1024, dev, seed=5)
    fn = lambda: eval_dan.detect_images(net, imgs, pyramid=False)
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 3
    for _ in range(n):
        out, num = fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print("%-60s %8.1f ms / image   %7.1f images/s   (batch %d, %s boxes out)" % ("eval_dan.detect_images, same passes, batched", dt / B * 1e3, B / dt, B, num.tolist()))
