"""The DAN context module (net/danet.py:842-918) as ONE autograd node over channel-slice views (ops._ContextBlock, round 4) against the same
block as ten separate convolutions + concat + add (the round-3 form, itself pinned against the oracle by tests/test_models_gpu.py and
tests/test_grad_parity_gpu.py): forward, input gradient and every variable's gradient; and the strided entry points of the C ABI against
their dense counterparts on contiguous copies of the same slices."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _block_outputs(fused, x, dy, seed, dev, C):
    from dan_amd.net import danet
    from dan_amd.net.variables import VariableStore
    vs = VariableStore(device=dev, seed=seed)
    bb = danet.VGG16Backbone("channels_last", variables=vs)
    bb.FUSED_CONTEXT_BLOCK = fused
    g = torch.Generator().manual_seed(seed + 1)
    xin = x.clone().requires_grad_(True)
    out = bb.se_inception_block(xin, "blk")
    with torch.no_grad():                                # non-zero biases (the reference initialises them to zero)
        for n, p in vs.named():
            if n.endswith("/bias"):
                p.copy_((0.1 * torch.randn(p.shape, generator=g)).to(dev))
    for _, p in vs.named():
        p.grad = None
    xin = x.clone().requires_grad_(True)
    out = bb.se_inception_block(xin, "blk")
    out.backward(dy)
    torch.cuda.synchronize()
    return out.detach().float().cpu(), xin.grad.float().cpu(), {n: p.grad.detach().float().cpu() for n, p in vs.named()}


@pytest.mark.parametrize("N,H,W,C", [(2, 40, 40, 256), (1, 67, 45, 256), (2, 12, 20, 512), (3, 5, 5, 1024), (16, 3, 3, 256)])
def test_fused_context_block_matches_the_separate_convolutions(N, H, W, C, dev):
    from dan_amd import ops
    g = torch.Generator().manual_seed(N * 1000 + H)
    x = torch.randn((N, H, W, C), generator=g).to(ops.ACT).to(dev)
    dy = torch.randn((N, H, W, C), generator=g).to(ops.ACT).to(dev)
    y0, dx0, g0 = _block_outputs(False, x, dy, 7, dev, C)
    y1, dx1, g1 = _block_outputs(True, x, dy, 7, dev, C)
    assert (y1 - y0).abs().max().item() <= 2.0 ** -6 * y0.abs().max().item()
    assert (dx1 - dx0).norm().item() <= 0.02 * dx0.norm().item()
    assert set(g0) == set(g1) and len(g0) == 20
    for n in g0:
        rel = (g1[n] - g0[n]).norm().item() / (g0[n].norm().item() + 1e-12)
        # (16-bit storage of different intermediates on the two routes - e.g. branch 2 rounds conv(x) instead of avg(x) - flips a few ReLU
        # decisions next to zero: the 64-element bias gradients feel single pixels most)
        # ... and branch 2 is where the two routes differ by construction (pool of the rounded conv output against conv of the rounded pool)
        assert rel <= (0.10 if ("branch2" in n or n.endswith("/bias")) else 0.05), (n, rel)


def test_fused_context_block_without_gradients_and_without_slots(dev):
    """16-bit inference (no_grad) takes the fused forward too; with ops.USE_SLOTS = False every gradient travels through autograd's own edges."""
    from dan_amd import ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn((2, 24, 24, 256), generator=g).to(ops.ACT).to(dev)
    dy = torch.randn((2, 24, 24, 256), generator=g).to(ops.ACT).to(dev)
    y0, dx0, g0 = _block_outputs(True, x, dy, 9, dev, 256)
    ops.USE_SLOTS = False
    try:
        y1, dx1, g1 = _block_outputs(True, x, dy, 9, dev, 256)
    finally:
        ops.USE_SLOTS = True
    assert torch.equal(y0, y1)
    assert (dx1 - dx0).norm().item() <= 0.01 * dx0.norm().item()
    for n in g0:
        assert (g1[n] - g0[n]).norm().item() <= 0.01 * g0[n].norm().item() + 1e-6, n


@pytest.mark.parametrize("kh,kw,cin,cout,H,W", [(1, 1, 256, 64, 40, 40), (3, 1, 64, 32, 40, 40), (1, 3, 64, 32, 33, 21), (3, 3, 64, 64, 40, 40), (1, 1, 256, 192, 40, 40),
                                                 (3, 3, 64, 64, 6, 6), (3, 3, 64, 64, 80, 80), (3, 3, 64, 64, 47, 93), (3, 3, 128, 64, 40, 64)])
def test_strided_entry_points_equal_the_dense_calls(kh, kw, cin, cout, H, W, dev):
    """danhip_conv2d_{fwd,bwd_data,bwd_weight}_strided on channel slices of wider tensors against the dense calls on contiguous copies of the
    same slices: forward (incl. the partial ReLU), data gradient (mask + accumulate) and weight gradient."""
    from dan_amd import _lib, ops
    N = 2
    g = torch.Generator().manual_seed(kh * 100 + cin + H)
    big_x = torch.randn((N, H, W, cin + 128), generator=g).to(ops.ACT).to(dev)
    xv = big_x[..., 64:64 + cin]
    w = (torch.randn((kh, kw, cin, cout), generator=g) / (kh * kw * cin) ** 0.5).to(dev)
    b = torch.randn((cout,), generator=g).to(dev)
    d = ops._desc(N, H, W, cin, cout, kh, kw, 1)
    wf, wb = ops.pack_conv_weight(d, w, need_bwd=True)
    big_y = torch.zeros((N, H, W, cout + 64), dtype=ops.ACT, device=dev)
    yv = big_y[..., 32:32 + cout]
    relu_ch = cout if cout != 192 else 128
    p = _lib.ConvPitch(big_x.shape[-1], big_y.shape[-1], 0)
    _lib.call("danhip_conv2d_fwd_strided", ctypes.byref(d), ops._vptr(xv), _lib.ptr(wf), _lib.ptr(b), ops._vptr(yv), 1, relu_ch, ctypes.byref(p), None, 0, _lib.stream())
    ref = torch.empty((N, H, W, cout), dtype=ops.ACT, device=dev)
    xc = xv.contiguous()                 # (named: a temporary's block would be handed to the next allocation while the call still reads it)
    _lib.call("danhip_conv2d_fwd", ctypes.byref(d), _lib.ptr(xc), _lib.ptr(wf), _lib.ptr(b), _lib.ptr(ref), _lib.BF16, 0, None, _lib.stream())
    want = ref.float()
    want[..., :relu_ch] = torch.relu(want[..., :relu_ch])
    torch.cuda.synchronize()
    tol = 2.0 ** -7 * want.abs().max().item() + 1e-3
    assert (yv.float() - want).abs().max().item() <= tol
    assert float(big_y[..., :32].abs().max()) == 0.0 and float(big_y[..., 32 + cout:].abs().max()) == 0.0      # nothing written outside the slice
    # ---- data gradient: dy is a slice, dx a slice, the mask a slice of a third tensor
    big_dy = torch.randn((N, H, W, cout + 64), generator=g).to(ops.ACT).to(dev)
    dyv = big_dy[..., 32:32 + cout]
    big_dx = torch.randn((N, H, W, cin + 128), generator=g).to(ops.ACT).to(dev)
    keep = big_dx.clone()
    dxv = big_dx[..., 64:64 + cin]
    p2 = _lib.ConvPitch(big_dy.shape[-1], big_dx.shape[-1], big_x.shape[-1])
    _lib.call("danhip_conv2d_bwd_data_strided", ctypes.byref(d), ops._vptr(dyv), _lib.ptr(wb), ops._vptr(xv), ops._vptr(dxv), 1, ctypes.byref(p2), None, 0,
              _lib.stream())
    ref_dx = keep[..., 64:64 + cin].contiguous()
    dyc = dyv.contiguous()
    _lib.call("danhip_conv2d_bwd_data", ctypes.byref(d), _lib.ptr(dyc), _lib.ptr(wb), _lib.ptr(xc), _lib.ptr(ref_dx), 1, _lib.stream())
    torch.cuda.synchronize()
    assert (dxv.float() - ref_dx.float()).abs().max().item() <= 2.0 ** -6 * ref_dx.float().abs().max().item() + 1e-2
    assert torch.equal(big_dx[..., :64], keep[..., :64]) and torch.equal(big_dx[..., 64 + cin:], keep[..., 64 + cin:])
    # ---- weight gradient
    dw = torch.zeros((kh, kw, cin, cout), dtype=torch.float32, device=dev)
    db = torch.zeros((cout,), dtype=torch.float32, device=dev)
    p3 = _lib.ConvPitch(big_x.shape[-1], big_dy.shape[-1], 0)
    _lib.call("danhip_conv2d_bwd_weight_strided", ctypes.byref(d), ops._vptr(xv), ops._vptr(dyv), _lib.ptr(dw), _lib.ptr(db), cin, ctypes.byref(p3), None, 0,
              _lib.stream())
    dw_ref = torch.zeros_like(dw)
    db_ref = torch.zeros_like(db)
    _lib.call("danhip_conv2d_bwd_weight", ctypes.byref(d), _lib.ptr(xc), _lib.ptr(dyc), _lib.ptr(dw_ref), _lib.ptr(db_ref), cin, _lib.stream())
    torch.cuda.synchronize()
    assert (dw - dw_ref).abs().max().item() <= 1e-3 * dw_ref.abs().max().item() + 1e-4
    assert (db - db_ref).abs().max().item() <= 1e-3 * db_ref.abs().max().item() + 1e-4
