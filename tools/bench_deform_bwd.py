import os, sys, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dan_amd._lib import call, ptr, stream
dev = torch.device("cuda:0")
N, H, W, C, dg = 16, 160, 160, 256, 4
g = torch.Generator().manual_seed(0)
x = torch.randn((N, H, W, C), generator=g).to(torch.bfloat16).to(dev)
scale = float(os.environ.get("OFF_SCALE", "0.5"))
off = (torch.randn((N, H, W, dg * 18), generator=g) * scale).to(torch.bfloat16).to(dev)
dS = torch.randn((N * H * W, 9 * C), generator=g).to(torch.bfloat16).to(dev)
dx = torch.empty_like(x); doff = torch.empty_like(off)
ws = torch.empty((N * H * W * C + 64,), dtype=torch.float32, device=dev)
def run():
    call("danhip_deform_sample_bwd", ptr(x), ptr(off), ptr(dS), ptr(dx), ptr(doff), N, H, W, C, 3, 3, 1, 1, dg, 0, ptr(ws), ws.numel() * 4, stream())
for _ in range(2): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): run()
e1.record(); torch.cuda.synchronize()
print("form=%s off_scale=%s  %.3f ms per call (zero + scatter + convert)" % (os.environ.get("DANHIP_DEFORM_BWD_FORM", "0"), scale, e0.elapsed_time(e1) / 5))
