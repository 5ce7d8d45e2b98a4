"""Parity of the deformable convolution (deformable im2col HIP kernel + MFMA GEMM, through custom_op.deform_conv_op)
against the oracle restatement of cpp/Deform (oracle/deform.py): forward, dX, dW, dOffset; the identity
"zero offsets == SAME conv" (custom_op.py:132 initial state); samples outside the image; the high-edge clamp."""
import pytest
import torch

from oracle import deform as OD
from oracle import tf_ops as T

pytestmark = pytest.mark.gpu


def _case(seed, N, H, W, C, Cout, dg, off_scale):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((N, H, W, C), generator=g).to(torch.bfloat16)
    w = (torch.randn((Cout, C, 3, 3), generator=g) / (9 * C) ** 0.5).to(torch.bfloat16).float()
    off = (torch.randn((N, H, W, dg * 18), generator=g) * off_scale).to(torch.bfloat16)
    return x, w, off


@pytest.mark.parametrize("seed,N,H,W,C,Cout,dg,off_scale", [(0, 1, 8, 8, 64, 64, 4, 0.0), (1, 2, 9, 7, 64, 32, 4, 1.5), (2, 1, 12, 12, 128, 64, 2, 4.0),
                                                            (3, 1, 6, 10, 256, 256, 4, 0.7), (4, 2, 20, 27, 128, 64, 2, 1.0), (5, 1, 17, 9, 64, 64, 1, 6.0)])
def test_deform_conv_forward_backward(seed, N, H, W, C, Cout, dg, off_scale, dev):
    from dan_amd.utility import custom_op
    x, w, off = _case(seed, N, H, W, C, Cout, dg, off_scale)
    xr, wr, offr = x.float(), w.clone(), off.float()
    ref = OD.deform_conv_forward(xr.permute(0, 3, 1, 2), wr, offr.permute(0, 3, 1, 2), 1, 1, dg).permute(0, 2, 3, 1)
    g = torch.Generator().manual_seed(100 + seed)
    dy = torch.randn(ref.shape, generator=g).to(torch.bfloat16)
    dxr, dwr, doffr = OD.deform_conv_backward(xr.permute(0, 3, 1, 2), wr, offr.permute(0, 3, 1, 2), dy.float().permute(0, 3, 1, 2), 1, 1, dg)
    xd = x.to(dev).requires_grad_(True)
    wd = w.to(dev).requires_grad_(True)
    od = off.to(dev).requires_grad_(True)
    y = custom_op.deform_conv_op(xd, wd, od, [1, 1, 1, 1], "SAME", [1, 1, 1, 1], 1, dg)
    y.backward(dy.to(dev))
    torch.cuda.synchronize()
    # forward: S is rounded to bf16 before the GEMM and y once more: 2 bf16 roundings
    assert (y.float().cpu() - ref).abs().max().item() <= 2.0 ** -6 * ref.abs().max().item() + 1e-3
    if off_scale == 0.0:                                # identity KAT: zero offsets == plain SAME convolution
        plain = T.conv2d_same(xr, wr.permute(2, 3, 1, 0).contiguous(), None, stride=1)
        assert (y.float().cpu() - plain).abs().max().item() <= 2.0 ** -7 * plain.abs().max().item() + 1e-3
    for name, got, want in (("dx", xd.grad.float().cpu(), dxr.permute(0, 2, 3, 1)), ("dw", wd.grad.cpu(), dwr),
                            ("doffset", od.grad.float().cpu(), doffr.permute(0, 2, 3, 1))):
        scale = want.abs().max().item() + 1e-6
        err = (got - want).abs().max().item()
        assert err <= 2.0 ** -5 * scale + 2e-3, (name, err, scale)      # dS (bf16) feeds both gradients: 3 roundings


def test_deform_conv_at_the_baseline_level_with_spread_offsets(dev):
    """The context module's largest instance (BASELINE.json configs[4]: 160 x 160 x 256 -> 256, 4 deformable groups), one image, offsets
    N(0, 2 px): most samples leave their cell, thousands leave the image — forward, dX, dW and dOffset against oracle/deform.py
    (the gather path had been checked at full size only with ~zero offsets)."""
    from dan_amd.utility import custom_op
    N, H, W, C, Cout, dg = 1, 160, 160, 256, 256, 4
    x, w, off = _case(160, N, H, W, C, Cout, dg, 2.0)
    xr, wr, offr = x.float().permute(0, 3, 1, 2), w.clone(), off.float().permute(0, 3, 1, 2)
    ref = OD.deform_conv_forward(xr, wr, offr, 1, 1, dg).permute(0, 2, 3, 1)
    g = torch.Generator().manual_seed(161)
    dy = torch.randn(ref.shape, generator=g).to(torch.bfloat16)
    dxr, dwr, doffr = OD.deform_conv_backward(xr, wr, offr, dy.float().permute(0, 3, 1, 2), 1, 1, dg)
    xd = x.to(dev).requires_grad_(True)
    wd = w.to(dev).requires_grad_(True)
    od = off.to(dev).requires_grad_(True)
    y = custom_op.deform_conv_op(xd, wd, od, [1, 1, 1, 1], "SAME", [1, 1, 1, 1], 1, dg)
    y.backward(dy.to(dev))
    torch.cuda.synchronize()
    assert (y.float().cpu() - ref).abs().max().item() <= 2.0 ** -6 * ref.abs().max().item() + 1e-3
    for name, got, want in (("dx", xd.grad.float().cpu(), dxr.permute(0, 2, 3, 1)), ("dw", wd.grad.cpu(), dwr),
                            ("doffset", od.grad.float().cpu(), doffr.permute(0, 2, 3, 1))):
        scale = want.abs().max().item() + 1e-6
        err = (got - want).abs().max().item()
        assert err <= 2.0 ** -5 * scale + 2e-3, (name, err, scale)
        rel = (got - want).norm().item() / (want.norm().item() + 1e-12)
        assert rel <= 0.02, (name, rel)


@pytest.mark.parametrize("seed,N,H,W,C,Cout,dg,off_scale", [(11, 1, 6, 10, 256, 256, 4, 0.7), (12, 3, 19, 23, 128, 128, 2, 2.5), (13, 2, 16, 16, 64, 64, 1, 6.0),
                                                            (14, 1, 40, 40, 256, 256, 4, 1.0), (15, 1, 5, 5, 256, 128, 4, 0.3)])
def test_fused_sampling_gemm_kernel(seed, N, H, W, C, Cout, dg, off_scale, dev):
    """csrc/deform_fused.hip (row a16: the deformable im2col feeds the GEMM through LDS): with gradients off no column buffer exists; its
    output equals the training forward (same kernel, column buffer also written) bit for bit and meets the oracle; the column buffer
    the training forward leaves behind equals the stand-alone sampling kernel's."""
    from dan_amd import ops
    from dan_amd._lib import lib
    x, w, off = _case(seed, N, H, W, C, Cout, dg, off_scale)
    assert lib().danhip_deform_conv_fused(N, H, W, C, Cout, 3, 3, 1, dg) == 1
    g = torch.Generator().manual_seed(seed)
    b = torch.randn(Cout, generator=g)
    ref = OD.deform_conv_forward(x.float().permute(0, 3, 1, 2), w, off.float().permute(0, 3, 1, 2), 1, 1, dg).permute(0, 2, 3, 1) + b
    w1 = w.permute(2, 3, 1, 0).reshape(1, 1, 9 * C, Cout).contiguous().to(dev)
    xd, od, bd = x.to(dev), off.to(dev), b.to(dev)
    torch.cuda.reset_peak_memory_stats()
    base = torch.cuda.memory_allocated()
    with torch.no_grad():
        y0 = ops.deform_conv(xd, w1, bd, od, 3, 3, deformable_group=dg, relu=False)
    torch.cuda.synchronize()
    if 9 * x.numel() * 2 > (4 << 20):                                   # (the packed filter, 1.2 MB, dwarfs the tiny cases' buffers)
        assert torch.cuda.max_memory_allocated() - base < 9 * x.numel() * 2, "inference allocated something as large as the column buffer"
    assert (y0.float().cpu() - ref).abs().max().item() <= 2.0 ** -6 * ref.abs().max().item() + 1e-3
    y1 = ops.deform_conv(xd.clone().requires_grad_(True), w1.clone().requires_grad_(True), bd, od, 3, 3, deformable_group=dg, relu=False)
    assert torch.equal(y0, y1.detach())
    col = y1.grad_fn.col
    assert col is not None
    S = ops.deform_sample(xd, od, 3, 3, deformable_group=dg)
    assert torch.equal(col[: S.numel() * 2].view(torch.bfloat16).view(S.shape), S)
    with torch.no_grad():                                               # ReLU epilogue
        yr = ops.deform_conv(xd, w1, bd, od, 3, 3, deformable_group=dg, relu=True)
    assert torch.equal(yr, torch.relu(y0))


def test_deform_conv_rejects_bad_arguments(dev):
    from dan_amd.utility import custom_op
    x, w, off = _case(0, 1, 4, 4, 64, 64, 4, 0.0)
    with pytest.raises(ValueError):
        custom_op.deform_conv_op(x.to(dev), w.to(dev), off.to(dev)[..., :70].contiguous(), [1, 1, 1, 1], "SAME", [1, 1, 1, 1], 1, 4)
    with pytest.raises(ValueError):
        custom_op.deform_conv_op(x.to(dev), w.to(dev), off.to(dev), [1, 1, 1, 1], "VALID", [1, 1, 1, 1], 1, 4)


def test_single_call_entry_points_equal_the_two_halves(dev):
    """danhip_deform_conv_{fwd,bwd} (DeformConvOp / DeformConvBackpropOp as one call each, workspace-resident im2col) against
    the separately exposed halves (deform_sample + 1x1 conv ops): bit-identical outputs and gradients, bias + fused ReLU on."""
    from dan_amd import ops
    from dan_amd._lib import DanhipError, call, lib, ptr, stream
    x, w, off = _case(7, 2, 11, 13, 128, 64, 2, 2.0)
    g = torch.Generator().manual_seed(8)
    b = torch.randn(64, generator=g)
    dy = torch.randn((2, 11, 13, 64), generator=g).to(torch.bfloat16).to(dev)
    res = []
    ops.USE_SPLITK = False                              # bit-identity: the separate conv op must run the same single-pass GEMM kernel as the library call
    for single in (True, False, "resample"):            # "resample": the reference's re-im2col in backward instead of the kept buffer
        ops.KEEP_DEFORM_COL = single != "resample"
        xd, od = x.to(dev).requires_grad_(True), off.to(dev).requires_grad_(True)
        w1 = w.permute(2, 3, 1, 0).reshape(1, 1, 9 * 128, 64).contiguous().to(dev).requires_grad_(True)
        bd = b.to(dev).requires_grad_(True)
        if single:
            y = ops.deform_conv(xd, w1, bd, od, 3, 3, deformable_group=2, relu=True)
        else:
            y = ops.conv2d(ops.deform_sample(xd, od, 3, 3, deformable_group=2), w1, bd, relu=True)
        y.backward(dy)
        res.append((y, xd.grad, od.grad, w1.grad, bd.grad))
    ops.KEEP_DEFORM_COL = True
    ops.USE_SPLITK = True
    for other in (res[1], res[2]):
        for a, c, name in zip(res[0], other, ("y", "dx", "doffset", "dw", "db")):
            if name in ("dw", "db", "dx"):        # fp32 atomics: order-dependent in the last bits
                assert torch.allclose(a.float(), c.float(), rtol=1e-2, atol=1e-3 * c.float().abs().max().item()), name
            else:
                assert torch.equal(a, c), name
    need = lib().danhip_deform_conv_workspace_bytes(2, 11, 13, 128, 3, 3, 1, 0)
    assert need >= 2 * 11 * 13 * 9 * 128 * 2
    ws = torch.empty(16, dtype=torch.uint8, device=dev)
    with pytest.raises(DanhipError):
        call("danhip_deform_conv_fwd", ptr(x.to(dev)), ptr(x.to(dev)), None, ptr(off.to(dev)), ptr(dy), 2, 11, 13, 128, 64, 3, 3, 1, 1, 2, 0, ptr(ws), 16,
             stream())


def _deform_block_outputs(use_slots, x, dy, off_std, dev):
    """The DAN-Deform context module (net/danet_deform.py:267-290: 1x1 down + ReLU -> offset conv | deformable 3x3 + ReLU -> 1x1 up + ReLU,
    + residual) on random variables with offsets of spread off_std: -> out, d input, {variable: gradient}."""
    from dan_amd import ops
    from dan_amd.net import danet_deform
    from dan_amd.net.variables import VariableStore
    vs = VariableStore(device=dev, seed=21)
    bb = danet_deform.VGG16Backbone("channels_last", variables=vs)
    with torch.no_grad():
        bb.se_inception_block(x, "blk")
        g = torch.Generator().manual_seed(22)
        for n, p in vs.named():
            if n.endswith("/bias"):
                p.copy_((0.1 * torch.randn(p.shape, generator=g)).to(dev))
            if n.endswith("deform_conv/conv2d/kernel"):          # the offset convolution (zero-initialised in the reference)
                p.copy_((off_std / 48.0 * torch.randn(p.shape, generator=g)).to(dev))
    ops.USE_SLOTS = use_slots
    try:
        xin = x.clone().requires_grad_(True)
        out = bb.se_inception_block(xin, "blk")
        out.backward(dy)
        torch.cuda.synchronize()
    finally:
        ops.USE_SLOTS = True
    return out.detach().float().cpu(), xin.grad.float().cpu(), {n: p.grad.detach().float().cpu() for n, p in vs.named()}


@pytest.mark.parametrize("N,H,W,off_std", [(2, 24, 24, 0.5), (1, 19, 33, 2.5)])
def test_deformable_backward_delivers_into_the_producers_slot(N, H, W, off_std, dev):
    """Round 4: the deformable convolution's input gradient goes straight into the slot of the layer below (danhip_deform_conv_bwd_deliver:
    times (x > 0) of the 1x1 'down' convolution's ReLU, added to the offset convolution's contribution) instead of through an autograd tensor
    that the producer clones, masks and adds.  Against the all-autograd route (ops.USE_SLOTS = False) through the same kernels - gather form
    (small offsets) and scatter form (large ones)."""
    from dan_amd import ops
    g = torch.Generator().manual_seed(N + H)
    x = torch.randn((N, H, W, 256), generator=g).to(ops.ACT).to(dev)
    dy = torch.randn((N, H, W, 256), generator=g).to(ops.ACT).to(dev)
    y0, dx0, g0 = _deform_block_outputs(False, x, dy, off_std, dev)
    y1, dx1, g1 = _deform_block_outputs(True, x, dy, off_std, dev)
    assert torch.equal(y0, y1)
    assert (dx1 - dx0).norm().item() <= 0.01 * dx0.norm().item()
    assert set(g0) == set(g1)
    for n in g0:
        assert (g1[n] - g0[n]).norm().item() <= 0.01 * g0[n].norm().item() + 1e-6, n
