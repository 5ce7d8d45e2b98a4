"""DeformConvOp forward at the context modules' shape: the fused sampling+GEMM kernel (csrc/deform_fused.hip) with / without the column
buffer against the two-kernel form (deform_sample + pointwise GEMM).  usage: python tools/bench_deform_fwd.py [H] [sigma]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dan_amd import ops

H = int(sys.argv[1]) if len(sys.argv) > 1 else 160
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
dev = torch.device("cuda:0")
N, C, Cout, dg = 16, 256, 256, 4
g = torch.Generator().manual_seed(0)
x = torch.randn((N, H, H, C), generator=g).to(torch.bfloat16).to(dev)
off = (torch.randn((N, H, H, dg * 18), generator=g) * sigma).to(torch.bfloat16).to(dev)
w1 = (torch.randn((1, 1, 9 * C, Cout), generator=g) / (9 * C) ** 0.5).to(dev)
b = torch.randn(Cout, generator=g).to(dev)
flops = 2.0 * N * H * H * 9 * C * Cout


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def fused_nocol():
    with torch.no_grad():
        return ops.deform_conv(x, w1, b, off, 3, 3, deformable_group=dg, relu=True)


wg = w1.clone().requires_grad_(True)


def fused_col():
    return ops.deform_conv(x, wg, b, off, 3, 3, deformable_group=dg, relu=True)


def two_kernels():
    with torch.no_grad():
        return ops.conv2d(ops.deform_sample(x, off, 3, 3, deformable_group=dg), w1, b, relu=True)


for name, fn in (("two kernels (sample + GEMM)", two_kernels), ("fused, no column buffer", fused_nocol), ("fused + column buffer", fused_col),
                 ("two kernels (sample + GEMM)", two_kernels), ("fused, no column buffer", fused_nocol)):
    ms = timeit(fn)
    print("%-32s %8.3f ms  %7.1f TFLOP/s" % (name, ms, flops / ms / 1e9))
