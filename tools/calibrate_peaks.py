#!/usr/bin/env python3
"""Calibrates the two rooflines on the box the bench runs on: a library bf16 GEMM (hipBLASLt through torch.matmul,
random N(0,1) operands) for the MFMA roofline, and fill / copy / read streams for the HBM one.  Prints one JSON object."""
import json
import sys

import torch

dev = torch.device("cuda:0")


def timed(f, it=20, warm=3):
    for _ in range(warm):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


out = {}
for n in (4096, 8192):
    a = torch.randn((n, n), device=dev, dtype=torch.float32).to(torch.bfloat16)
    b = torch.randn((n, n), device=dev, dtype=torch.float32).to(torch.bfloat16)
    ms = timed(lambda: torch.matmul(a, b), it=10)
    out["gemm_bf16_%d_tflops" % n] = round(2.0 * n ** 3 / ms / 1e9, 1)
n = 16 * 640 * 640 * 64                      # the conv1_2 activation: 839 MB of bf16
x = torch.randn(n, device=dev, dtype=torch.float32).to(torch.bfloat16)
y = torch.empty_like(x)
out["fill_839MB_TBps"] = round(n * 2 / timed(lambda: y.zero_()) / 1e9, 2)
out["copy_839MB_TBps_read_plus_write"] = round(2 * n * 2 / timed(lambda: y.copy_(x)) / 1e9, 2)
out["read_839MB_TBps"] = round(n * 2 / timed(lambda: x.sum()) / 1e9, 2)
print(json.dumps(out))
