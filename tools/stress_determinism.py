"""Race hunt for the one-barrier kernels: forward / data gradient (bit mask) of the 3x3 halo kernels and the slab-form weight gradient are
free of atomics, so repeated launches on the same inputs must be BIT-identical.  A hand-off race (a ring slot or patch buffer rewritten
before its last reader is done) would show as a run that differs.  The reference of every shape is computed with the SECOND barrier per step
switched back on (options halo_b2 / wgrad_b2: the round-2 form), the repeats run in the default one-barrier form while a second stream keeps
launching weight gradients (the training step's own concurrency: timing noise).  usage: python tools/stress_determinism.py [repeats]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dan_amd import ops
from dan_amd._lib import BF16, call, lib, ptr, stream

REPS = int(sys.argv[1]) if len(sys.argv) > 1 else 150
SHAPES = [("conv2_1", 16, 320, 320, 64, 128), ("conv2_2", 16, 320, 320, 128, 128), ("conv3_1", 16, 160, 160, 128, 256), ("conv3_2", 16, 160, 160, 256, 256),
          ("conv4_2", 16, 80, 80, 512, 512), ("ragged", 3, 203, 171, 192, 136)]
dev = torch.device("cuda:0")
bad = 0
for name, N, H, W, Cin, Cout in SHAPES:
    g = torch.Generator(device="cpu").manual_seed(1)
    x = torch.randn((N, H, W, Cin), generator=g).to(ops.ACT).to(dev)
    w = (torch.randn((3, 3, Cin, Cout), generator=g) / (9 * Cin) ** 0.5).to(torch.bfloat16).float().to(dev)
    b = torch.randn((Cout,), generator=g).to(dev)
    d = ops._desc(N, H, W, Cin, Cout, 3, 3, 1)
    wf, wb = ops.pack_conv_weight(d, w, need_bwd=True)
    co8 = (Cout + 7) // 8 * 8
    dy = torch.zeros((N, H, W, co8), dtype=ops.ACT)
    dy[..., :Cout] = torch.randn((N, H, W, Cout), generator=g).to(ops.ACT)
    dy = dy.to(dev)
    bits = torch.empty((N * H * W, Cin // 8), dtype=torch.uint8, device=dev)
    call("danhip_relu_bits", ptr(x), ptr(bits), N * H * W, Cin, stream())
    takes_bits = bool(lib().danhip_conv2d_bwd_data_takes_bits(ctypes.byref(d)))
    ref = {}
    side = torch.cuda.Stream()
    dw = torch.zeros((3, 3, Cin, Cout), dtype=torch.float32, device=dev)
    db = torch.zeros((Cout,), dtype=torch.float32, device=dev)
    for rep in range(REPS + 1):
        lib().danhip_set_option(b"halo_b2", 1 if rep == 0 else 0)
        if rep > 0:
            with torch.cuda.stream(side):
                for _ in range(2):
                    call("danhip_conv2d_bwd_weight", ctypes.byref(d), ptr(x), ptr(dy), ptr(dw), ptr(db), Cin, stream())
        y = torch.empty((N, H, W, Cout), dtype=ops.ACT, device=dev)
        dx = torch.empty_like(x)
        call("danhip_conv2d_fwd", ctypes.byref(d), ptr(x), ptr(wf), ptr(b), ptr(y), BF16, 1, None, stream())
        if takes_bits:
            call("danhip_conv2d_bwd_data_bits", ctypes.byref(d), ptr(dy), ptr(wb), ptr(bits), ptr(dx), 0, stream())
        else:
            call("danhip_conv2d_bwd_data", ctypes.byref(d), ptr(dy), ptr(wb), ptr(x), ptr(dx), 0, stream())
        torch.cuda.synchronize()
        for k, t in (("fwd", y), ("dgrad", dx)):
            if rep == 0:
                ref[k] = t.clone()
            elif not torch.equal(t, ref[k]):
                bad += 1
                print("MISMATCH %s %s rep %d: %d elements differ" % (name, k, rep, int((t != ref[k]).sum().item())))
    print("%-8s %d repeats: forward and data gradient bit-identical" % (name, REPS) if not bad else "%s: %d mismatching runs so far" % (name, bad))

# ---- the weight gradients (conv_wgrad_rows.hip for the 3x3 shapes, conv_wgrad_pw.hip for the 1x1 ones) in their atomic-free SLAB form
# (wgrad_slab = 2: partial tiles as plain stores + a fixed-order combine pass), so dW / db are bit-reproducible: reference with the second
# barrier per K-step (wgrad_b2 = 1), repeats in the default one-barrier form into a zeroed dw, with forward launches on a second stream
WSHAPES = [("conv2_2", 16, 320, 320, 128, 128, 3), ("conv3_2", 16, 160, 160, 256, 256, 3), ("conv4_2", 16, 80, 80, 512, 512, 3), ("conv2_1", 16, 320, 320, 64, 128, 3),
           ("lateral", 16, 160, 160, 256, 256, 1), ("up", 16, 80, 80, 512, 256, 1), ("deformK", 4, 160, 160, 2304, 256, 1)]
lib().danhip_set_option(b"wgrad_slab", 2)
try:
    for name, N, H, W, Cin, Cout, k in WSHAPES:
        g = torch.Generator(device="cpu").manual_seed(2)
        x = torch.randn((N, H, W, Cin), generator=g).to(ops.ACT).to(dev)
        dy = torch.randn((N, H, W, Cout), generator=g).to(ops.ACT).to(dev)
        d = ops._desc(N, H, W, Cin, Cout, k, k, 1)
        nws = lib().danhip_conv2d_bwd_weight_workspace_bytes(ctypes.byref(d))
        if not nws:
            print("%-8s: the library offers no slab form for this shape - skipped" % name)
            continue
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        w0 = (torch.randn((k, k, Cin, Cout), generator=g) / (k * k * Cin) ** 0.5).to(dev)
        wf, _ = ops.pack_conv_weight(d, w0, need_bwd=False)
        b0 = torch.zeros((Cout,), device=dev)
        side = torch.cuda.Stream()
        ref = None
        for rep in range(REPS + 1):
            lib().danhip_set_option(b"wgrad_b2", 1 if rep == 0 else 0)
            if rep > 0:
                with torch.cuda.stream(side):
                    y = torch.empty((N, H, W, Cout), dtype=ops.ACT, device=dev)
                    call("danhip_conv2d_fwd", ctypes.byref(d), ptr(x), ptr(wf), ptr(b0), ptr(y), BF16, 1, None, stream())
            dw = torch.zeros((k, k, Cin, Cout), dtype=torch.float32, device=dev)
            db = torch.zeros((Cout,), dtype=torch.float32, device=dev)
            call("danhip_conv2d_bwd_weight_ws", ctypes.byref(d), ptr(x), ptr(dy), ptr(dw), ptr(db), Cin, ptr(ws), nws, stream())
            torch.cuda.synchronize()
            if rep == 0:
                ref = (dw.clone(), db.clone())
            elif not torch.equal(dw, ref[0]) or (db - ref[1]).abs().max().item() > 1e-5 * ref[1].abs().max().item():
                bad += 1          # (db stays a few fp32 atomics per channel in both forms: compared up to summation order)
                print("MISMATCH %s wgrad rep %d: %d dW elements differ, db max |diff| %.3g" % (name, rep, int((dw != ref[0]).sum().item()), (db - ref[1]).abs().max().item()))
        print("%-8s %d repeats: weight gradient (slab form) bit-identical" % (name, REPS) if not bad else "%s: %d mismatching runs so far" % (name, bad))
finally:
    lib().danhip_set_option(b"wgrad_slab", 1)
    lib().danhip_set_option(b"wgrad_b2", 0)
sys.exit(1 if bad else 0)
