#!/usr/bin/env python3
"""Per-layer microbenchmark of the libdanhip convolution kernels (through the C ABI), with a quick numerical check.

    python tools/bench_conv.py [--set s3fd|pb|small] [--iters 20] [--check] [--which fwd,dgrad,wgrad]

Reports, per shape, the mean launch duration (HIP events on the launch stream) and algorithmic TFLOP/s
(2*N*Ho*Wo*Cin*Cout*kh*kw per launch).  Random N(0,1) bf16 data (never zeros: DVFS, cdna_hip_programming.md rule 25).
"""
import argparse
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

from dan_amd import ops
from dan_amd._lib import BF16, DanhipError, call, lib, ptr, stream

# (name, N, H, W, Cin, Cout, k, stride)
S3FD = [
    ("conv1_1", 16, 640, 640, 8, 64, 3, 1), ("conv1_2", 16, 640, 640, 64, 64, 3, 1), ("conv2_1", 16, 320, 320, 64, 128, 3, 1), ("conv2_2", 16, 320, 320, 128, 128, 3, 1),
    ("conv3_1", 16, 160, 160, 128, 256, 3, 1), ("conv3_2", 16, 160, 160, 256, 256, 3, 1), ("conv4_1", 16, 80, 80, 256, 512, 3, 1),
    ("conv4_2", 16, 80, 80, 512, 512, 3, 1), ("conv5_1", 16, 40, 40, 512, 512, 3, 1), ("fc6", 16, 20, 20, 512, 1024, 3, 1),
    ("fc7", 16, 20, 20, 1024, 1024, 1, 1),
]
PB = [
    ("cpm160_a", 4, 160, 160, 256, 1024, 3, 1), ("cpm160_b", 4, 160, 160, 1024, 256, 3, 1), ("cpm80_a", 4, 80, 80, 512, 1024, 3, 1),
    ("lat160", 16, 160, 160, 256, 256, 1, 1), ("dfm160", 16, 160, 160, 2304, 256, 1, 1), ("dfm80", 16, 80, 80, 2304, 256, 1, 1),
    ("ctx160", 16, 160, 160, 256, 128, 3, 1), ("up1024", 16, 160, 160, 256, 1024, 1, 1),
]
# 1024x1024 inputs (BASELINE.json configs[3..4]) and sizes that are multiples of nothing.  NB: --check uses torch's fp32 NCHW conv as the
# yardstick, which itself breaks down on b1_2 (4.3 GB operand); tests/test_conv_gpu.py covers >= 2 GiB activations by self-consistency
BIG = [("b1_2", 16, 1024, 1024, 64, 64, 3, 1), ("b2_2", 8, 512, 512, 128, 128, 3, 1), ("b3_2", 8, 256, 256, 256, 256, 3, 1),
       ("b4_2", 8, 128, 128, 512, 512, 3, 1), ("b5_1", 8, 64, 64, 512, 512, 3, 1), ("bfc6", 8, 32, 32, 512, 1024, 3, 1),
       ("odd1", 3, 203, 331, 64, 64, 3, 1), ("odd2", 3, 102, 166, 128, 128, 3, 1), ("odd3", 2, 51, 83, 256, 256, 3, 1),
       ("odd4", 2, 26, 42, 512, 512, 3, 1), ("oddh", 2, 102, 166, 256, 8, 3, 1)]
# the rest of the S3FD graph: extra layers (net/sfd_net.py:146-156) and the six loc+cls heads (Cout = 4 + 2, conv3_3 head with max-out: 4 + 4)
TAIL = [("conv6_1", 16, 20, 20, 1024, 256, 1, 1), ("conv6_2", 16, 20, 20, 256, 512, 3, 2), ("conv7_1", 16, 10, 10, 512, 128, 1, 1),
        ("conv7_2", 16, 10, 10, 128, 256, 3, 2), ("head3_3", 16, 160, 160, 256, 8, 3, 1), ("head4_3", 16, 80, 80, 512, 6, 3, 1),
        ("head5_3", 16, 40, 40, 512, 6, 3, 1), ("headfc7", 16, 20, 20, 1024, 6, 3, 1), ("head6_2", 16, 10, 10, 512, 6, 3, 1),
        ("head7_2", 16, 5, 5, 256, 6, 3, 1)]
# DAN context module V1 at the 160x160 level (net/danet.py:842-918): branch 1x1s, 3x1 / 1x3 (as 3x3 here: kh = kw in this tool), residual 1x1
DANB = [("b_1x1", 16, 160, 160, 256, 64, 1, 1), ("b4_3x3", 16, 160, 160, 64, 64, 3, 1), ("res_1x1", 16, 160, 160, 192, 256, 1, 1),
        ("s2_mix_a", 16, 160, 160, 256, 88, 1, 1), ("s2_mix_b", 16, 160, 160, 256, 176, 1, 1), ("b_1x1_80", 16, 80, 80, 512, 64, 1, 1),
        ("res_80", 16, 80, 80, 192, 512, 1, 1)]
SMALL = [("s64", 2, 32, 64, 64, 64, 3, 1), ("s128", 2, 24, 40, 128, 128, 3, 1), ("s256", 1, 48, 48, 256, 256, 3, 1)]


def run(shape, iters, which, check):
    name, N, H, W, Cin, Cout, k, s = shape
    dev = torch.device("cuda:0")
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn((N, H, W, Cin), generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn((k, k, Cin, Cout), generator=g) / (k * k * Cin) ** 0.5).to(torch.bfloat16).float().to(dev)
    b = torch.randn((Cout,), generator=g).to(dev)
    d = ops._desc(N, H, W, Cin, Cout, k, k, s)
    wf, wb = ops.pack_conv_weight(d, w, need_bwd=True)
    y = torch.empty((N, d.Ho, d.Wo, Cout), dtype=torch.bfloat16, device=dev)
    co8 = (Cout + 7) // 8 * 8                          # the gradient kernels take dY channel-padded to a multiple of 8 (zeros in the pad lanes)
    dy = torch.zeros((N, d.Ho, d.Wo, co8), dtype=torch.bfloat16)
    dy[..., :Cout] = torch.randn((N, d.Ho, d.Wo, Cout), generator=g).to(torch.bfloat16)
    dy = dy.to(dev)
    dx = torch.empty_like(x)
    cin_real = 3 if Cin == 8 else Cin                  # (the first layer: the image's 3 channels padded to 8)
    dw = torch.zeros((k, k, cin_real, Cout), dtype=torch.float32, device=dev)
    db = torch.zeros((Cout,), dtype=torch.float32, device=dev)
    flops = 2.0 * N * d.Ho * d.Wo * Cin * Cout * k * k

    nwf, nwd = lib().danhip_conv2d_workspace_bytes(ctypes.byref(d), 0), lib().danhip_conv2d_workspace_bytes(ctypes.byref(d), 1)
    wsf = torch.empty(max(nwf, nwd, 16), dtype=torch.uint8, device=dev)

    def fwd():
        call("danhip_conv2d_fwd_ws", ctypes.byref(d), ptr(x), ptr(wf), ptr(b), ptr(y), BF16, 1, None, ptr(wsf) if nwf else None, nwf, stream())

    def dgrad():
        call("danhip_conv2d_bwd_data_ws", ctypes.byref(d), ptr(dy), ptr(wb), ptr(x), ptr(dx), 0, ptr(wsf) if nwd else None, nwd, stream())

    def dgrad_nomask():
        call("danhip_conv2d_bwd_data", ctypes.byref(d), ptr(dy), ptr(wb), None, ptr(dx), 0, stream())

    bits = torch.empty((N * H * W, Cin // 8), dtype=torch.uint8, device=dev)

    def relu_bits():
        call("danhip_relu_bits", ptr(x), ptr(bits), N * H * W, Cin, stream())

    def dgrad_bits():
        call("danhip_conv2d_bwd_data_bits", ctypes.byref(d), ptr(dy), ptr(wb), ptr(bits), ptr(dx), 0, stream())

    def dgrad_acc():
        call("danhip_conv2d_bwd_data", ctypes.byref(d), ptr(dy), ptr(wb), ptr(x), ptr(dx), 1, stream())

    # conv1_2's data gradient with conv1_1's weight / bias gradient folded in (64 -> 64 only): dX never stored
    img8 = torch.zeros((N, H, W, 8), dtype=torch.bfloat16)
    img8[..., :3] = torch.randn((N, H, W, 3), generator=g).to(torch.bfloat16)
    img8 = img8.to(dev)
    dw8 = torch.zeros((3, 3, 3, 64), dtype=torch.float32, device=dev)
    db8 = torch.zeros((64,), dtype=torch.float32, device=dev)

    def dgrad_first():
        call("danhip_conv2d_bwd_data_bits_first", ctypes.byref(d), ptr(dy), ptr(wb), ptr(bits), ptr(img8), 3, ptr(dw8), ptr(db8), stream())

    nws = lib().danhip_conv2d_bwd_weight_workspace_bytes(ctypes.byref(d))
    ws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev)

    def wgrad():
        call("danhip_conv2d_bwd_weight_ws", ctypes.byref(d), ptr(x), ptr(dy), ptr(dw), ptr(db), cin_real, ptr(ws) if nws else None, nws, stream())

    fns = {"fwd": fwd, "dgrad": dgrad, "wgrad": wgrad, "dgrad_nomask": dgrad_nomask, "dgrad_acc": dgrad_acc, "dgrad_bits": dgrad_bits,
           "relu_bits": relu_bits, "dgrad_first": dgrad_first}
    out = []
    for wname in which:
        fn = fns[wname]
        try:
            for _ in range(3):
                fn()
        except DanhipError:                            # (a form this shape's kernel does not take, e.g. the bit-mask data gradient on 64-wide tiles)
            continue
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        label = ""
        if wname not in ("wgrad", "relu_bits"):
            label = lib().danhip_conv_kernel_label(ctypes.byref(d), 0 if wname == "fwd" else (1 if wname == "dgrad_nomask" else 5)).decode()
        out.append((wname, ms, flops / ms / 1e9, label))
    errs = {}
    if check:
        import torch.nn.functional as F
        pt = max((d.Ho - 1) * s + k - H, 0)
        pl = max((d.Wo - 1) * s + k - W, 0)
        xn = F.pad(x.float().permute(0, 3, 1, 2), (pl // 2, pl - pl // 2, pt // 2, pt - pt // 2))
        xn.requires_grad_(True)
        wr = w.clone().requires_grad_(True)
        ref = F.conv2d(xn, wr.permute(3, 2, 0, 1), b, stride=s)
        refr = torch.relu(ref).permute(0, 2, 3, 1)
        fwd()
        errs["fwd"] = ((y.float() - refr).abs().max() / refr.abs().max()).item()
        ref.backward(dy[..., :Cout].float().permute(0, 3, 1, 2))
        dgrad()
        gx = xn.grad[:, :, pt // 2:pt // 2 + H, pl // 2:pl // 2 + W].permute(0, 2, 3, 1) * (x.float() > 0)
        errs["dgrad"] = ((dx.float() - gx).abs().max() / gx.abs().max()).item()
        dw.zero_(); db.zero_()
        wgrad()
        errs["wgrad"] = ((dw - wr.grad[:, :, :cin_real]).abs().max() / wr.grad.abs().max()).item()
        dbr = dy[..., :Cout].float().sum((0, 1, 2))
        errs["db"] = ((db - dbr).abs().max() / dbr.abs().max()).item()
    return out, errs


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--set", default="s3fd")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--which", default="fwd,dgrad,wgrad")
    ap.add_argument("--only", default="")
    ap.add_argument("--batch", type=int, default=0, help="override N of every shape (the per-rank shapes of the strong-scaling series)")
    args = ap.parse_args()
    shapes = {"s3fd": S3FD, "pb": PB, "small": SMALL, "big": BIG, "tail": TAIL, "dan": DANB}[args.set]
    if args.only:
        shapes = [s for s in shapes if s[0] in args.only.split(",")]
    if args.batch:
        shapes = [(s[0], args.batch) + tuple(s[2:]) for s in shapes]
    which = args.which.split(",")
    tot = {w: [0.0, 0.0] for w in which}
    for sh in shapes:
        res, errs = run(sh, args.iters, which, args.check)
        for wname, ms, tf, label in res:
            tot[wname][0] += ms
            tot[wname][1] += tf * ms
            e = ("  relerr=%.2e" % errs[wname]) if wname in errs else ""
            print("%-10s %-6s %8.3f ms %8.1f TFLOP/s  %s%s" % (sh[0], wname, ms, tf, label, e), flush=True)
        if "db" in errs:
            print("%-10s db relerr=%.2e" % (sh[0], errs["db"]))
    for wname, (ms, tfms) in tot.items():
        print("TOTAL %-6s %8.3f ms  %8.1f TFLOP/s" % (wname, ms, tfms / max(ms, 1e-9)))


if __name__ == "__main__":
    main()
