"""Does the deformable backward's sampling part run faster when dS is produced and consumed one image at a time (118 MB per image at
160 x 160 x 256: inside the 256 MiB Infinity Cache) than over the whole batch (1.89 GB)?  The producer is emulated by a device copy."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dan_amd._lib import call, ptr, stream
dev = torch.device("cuda:0")
N, H, W, C, dg = 16, 160, 160, 256, 4
g = torch.Generator().manual_seed(0)
x = torch.randn((N, H, W, C), generator=g).to(torch.bfloat16).to(dev)
scale = float(os.environ.get("OFF_SCALE", "0.3"))
off = (torch.randn((N, H, W, dg * 18), generator=g) * scale).to(torch.bfloat16).to(dev)
src = torch.randn((N, H * W, 9 * C), generator=g).to(torch.bfloat16).to(dev)
dS = torch.empty_like(src)
dx = torch.empty_like(x); doff = torch.empty_like(off)
ws = torch.empty((N * H * W * C + 64 * N,), dtype=torch.float32, device=dev)


def whole():
    dS.copy_(src)
    call("danhip_deform_sample_bwd", ptr(x), ptr(off), ptr(dS), ptr(dx), ptr(doff), N, H, W, C, 3, 3, 1, 1, dg, 0, ptr(ws), (N * H * W * C + 64) * 4, stream())


def chunked(k):
    per = H * W * C + 64
    for n in range(0, N, k):
        dS[n:n + k].copy_(src[n:n + k])
        call("danhip_deform_sample_bwd", ptr(x[n:n + k]), ptr(off[n:n + k]), ptr(dS[n:n + k]), ptr(dx[n:n + k]), ptr(doff[n:n + k]), k, H, W, C, 3, 3, 1, 1, dg, 0,
             ptr(ws[n * per:]), (k * H * W * C + 64) * 4, stream())


def copy_only():
    dS.copy_(src)


def timeit(fn, it=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


for name, fn in (("copy only", copy_only), ("whole batch", whole), ("1 image", lambda: chunked(1)), ("2 images", lambda: chunked(2)), ("4 images", lambda: chunked(4)),
                 ("whole batch", whole), ("1 image", lambda: chunked(1))):
    print("%-12s %.3f ms" % (name, timeit(fn)))
