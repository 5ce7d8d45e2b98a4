"""DAN's stage-2 input mix (net/danet.py:944-950: concat([relu(conv1x1(stop_gradient(stage1), C // 3)), relu(conv1x1(feature, C - C // 3))]))
as ONE streaming GEMM over the never-written channel concatenation with a block-diagonal kernel (ops._ConcatMix, danhip_conv2d_fwd_concat2,
round 4) against the two ragged convolutions + concat it replaces (themselves pinned against the oracle by tests/test_models_gpu.py and
tests/test_grad_parity_gpu.py), and the C entry point against the fp32 product."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("N,H,W,C1,C2,Co,relu", [(2, 48, 48, 256, 256, 256, 1), (1, 40, 56, 512, 512, 512, 1), (3, 33, 35, 64, 64, 64, 0), (1, 47, 61, 128, 128, 192, 1),
                                                   (2, 40, 40, 1024, 1024, 1024, 1)])
def test_concat2_entry_point_against_the_fp32_product(N, H, W, C1, C2, Co, relu, dev):
    from dan_amd import _lib, ops
    g = torch.Generator().manual_seed(C1 + Co + H)
    a = torch.randn((N, H, W, C1), generator=g).to(ops.ACT)
    f = torch.randn((N, H, W, C2), generator=g).to(ops.ACT)
    w = (torch.randn((1, 1, C1 + C2, Co), generator=g) / (C1 + C2) ** 0.5).to(ops.ACT).float()
    b = torch.randn((Co,), generator=g)
    d = ops._desc(N, H, W, C1 + C2, Co, 1, 1, 1)
    assert _lib.lib().danhip_conv2d_fwd_concat2_supported(ctypes.byref(d), C1, C1) == 1
    wf, _ = ops.pack_conv_weight(d, w.to(dev), need_bwd=False)
    ad, fd, bd = a.to(dev), f.to(dev), b.to(dev)
    y = torch.full((N, H, W, Co), 7.0, dtype=ops.ACT, device=dev)
    _lib.call("danhip_conv2d_fwd_concat2", ctypes.byref(d), _lib.ptr(ad), _lib.ptr(fd), C1, C1, _lib.ptr(wf), _lib.ptr(bd), _lib.ptr(y), relu, _lib.stream())
    torch.cuda.synchronize()
    ref = torch.cat([a, f], dim=-1).float().reshape(-1, C1 + C2) @ w.reshape(C1 + C2, Co) + b
    if relu:
        ref = ref.clamp_min(0.0)
    ref = ref.reshape(N, H, W, Co)
    err = (y.float().cpu() - ref).abs().max().item()
    assert err <= 2.0 ** -7 * ref.abs().max().item() + 1e-2, err
    # the same product through the ordinary call on the materialised concatenation: same kernel family, same K order -> bit-identical
    x = torch.cat([ad, fd], dim=-1).contiguous()
    y2 = torch.empty_like(y)
    _lib.call("danhip_conv2d_fwd", ctypes.byref(d), _lib.ptr(x), _lib.ptr(wf), _lib.ptr(bd), _lib.ptr(y2), _lib.BF16, relu, None, _lib.stream())
    torch.cuda.synchronize()
    assert torch.equal(y, y2)


def test_concat2_refuses_what_the_streaming_gemm_does_not_take(dev):
    from dan_amd import _lib, ops
    d = ops._desc(1, 8, 8, 512, 256, 1, 1, 1)                 # 64 pixels: below the streaming kernel's floor
    assert _lib.lib().danhip_conv2d_fwd_concat2_supported(ctypes.byref(d), 256, 256) == 0
    d = ops._desc(2, 48, 48, 512, 256, 1, 1, 1)
    assert _lib.lib().danhip_conv2d_fwd_concat2_supported(ctypes.byref(d), 200, 256) == 0       # c1 not a multiple of 64
    assert _lib.lib().danhip_conv2d_fwd_concat2_supported(ctypes.byref(d), 256, 200) == 0       # pitch below the channel count
    d3 = ops._desc(2, 48, 48, 512, 256, 3, 3, 1)
    assert _lib.lib().danhip_conv2d_fwd_concat2_supported(ctypes.byref(d3), 256, 256) == 0      # 1x1 only
    x = torch.zeros((2, 48, 48, 256), dtype=ops.ACT, device=dev)
    with pytest.raises(_lib.DanhipError):
        _lib.call("danhip_conv2d_fwd_concat2", ctypes.byref(d3), _lib.ptr(x), _lib.ptr(x), 256, 256, _lib.ptr(x), None, _lib.ptr(x), 1, _lib.stream())


def _mix_outputs(fused, stage1, f, dy, seed, dev):
    from dan_amd.net import danet
    from dan_amd.net.variables import VariableStore
    vs = VariableStore(device=dev, seed=seed)
    bb = danet.VGG16Backbone("channels_last", variables=vs)
    bb.FUSED_STAGE2_MIX = fused
    bb.se_inception_block = lambda x, name=None: x            # the mix alone (the context block behind it has its own test)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():
        bb.get_features_stage2([stage1], [f])
        for n, p in vs.named():                               # non-zero biases (the reference initialises them to zero)
            if n.endswith("/bias"):
                p.copy_((0.1 * torch.randn(p.shape, generator=g)).to(dev))
    fin = f.clone().requires_grad_(True)
    s1 = stage1.clone().requires_grad_(True)
    (out,) = bb.get_features_stage2([s1], [fin])
    out.backward(dy)
    torch.cuda.synchronize()
    assert s1.grad is None                                    # stop_gradient (net/danet.py:945)
    return out.detach().float().cpu(), fin.grad.float().cpu(), {n: p.grad.detach().float().cpu() for n, p in vs.named()}


@pytest.mark.parametrize("N,H,W,C", [(2, 40, 40, 256), (1, 67, 45, 256), (2, 40, 40, 512), (3, 5, 5, 1024), (16, 3, 3, 256)])
def test_fused_stage2_mix_matches_the_two_ragged_convolutions(N, H, W, C, dev):
    """Forward, the feature map's gradient and the four variables' gradients; the small maps take the concatenating fallback of the same node."""
    from dan_amd import ops
    g = torch.Generator().manual_seed(N * 100 + W)
    s1 = torch.randn((N, H, W, C), generator=g).to(ops.ACT).to(dev)
    f = torch.randn((N, H, W, C), generator=g).to(ops.ACT).to(dev)
    dy = torch.randn((N, H, W, C), generator=g).to(ops.ACT).to(dev)
    y0, df0, g0 = _mix_outputs(False, s1, f, dy, 11, dev)
    y1, df1, g1 = _mix_outputs(True, s1, f, dy, 11, dev)
    assert (y1 - y0).abs().max().item() <= 2.0 ** -6 * y0.abs().max().item()
    assert (df1 - df0).norm().item() <= 0.02 * df0.norm().item()
    assert set(g0) == set(g1) and len(g0) == 4
    for n in g0:
        assert g0[n].shape == g1[n].shape
        rel = (g1[n] - g0[n]).norm().item() / (g0[n].norm().item() + 1e-12)
        assert rel <= 0.03, (n, rel)


def test_fused_stage2_mix_in_the_trainer_layout(dev):
    """With the trainer's flat buffers the two kernels are the diagonal blocks of ONE [1, 1, 2C, C] block (FlatParams "blockdiag"): gradients
    land in the flat gradient buffer, its off-diagonal blocks stay exactly zero, and the values match the plain-autograd route."""
    from dan_amd import ops
    from dan_amd.net import danet
    from dan_amd.net.variables import VariableStore
    from dan_amd.trainer import FlatParams
    N, H, W, C = 2, 40, 40, 256
    c3 = C // 3
    g = torch.Generator().manual_seed(5)
    s1 = torch.randn((N, H, W, C), generator=g).to(ops.ACT).to(dev)
    f = torch.randn((N, H, W, C), generator=g).to(ops.ACT).to(dev)
    dy = torch.randn((N, H, W, C), generator=g).to(ops.ACT).to(dev)
    y0, df0, g0 = _mix_outputs(True, s1, f, dy, 11, dev)
    vs = VariableStore(device=dev, seed=11)
    bb = danet.VGG16Backbone("channels_last", variables=vs)
    bb.se_inception_block = lambda x, name=None: x
    with torch.no_grad():
        bb.get_features_stage2([s1], [f])
        gg = torch.Generator().manual_seed(12)
        for n, p in vs.named():
            if n.endswith("/bias"):
                p.copy_((0.1 * torch.randn(p.shape, generator=gg)).to(dev))
    flat = FlatParams(vs)
    names = [n for n, _ in vs.named()]
    key = (names[0], names[2])
    assert key in vs.fused and tuple(vs.fused[key].shape) == (1, 1, 2 * C, C)
    wv = vs.fused[key]
    assert wv[0, 0, :C, c3:].abs().max().item() == 0.0 and wv[0, 0, C:, :c3].abs().max().item() == 0.0
    flat.zero_grad()
    fin = f.clone().requires_grad_(True)
    (out,) = bb.get_features_stage2([s1], [fin])
    out.backward(dy)
    torch.cuda.synchronize()
    assert (out.detach().float().cpu() - y0).abs().max().item() <= 2.0 ** -6 * y0.abs().max().item()
    assert (fin.grad.float().cpu() - df0).norm().item() <= 0.01 * df0.norm().item()
    gblock = wv._danhip_grad
    assert gblock[0, 0, :C, c3:].abs().max().item() == 0.0 and gblock[0, 0, C:, :c3].abs().max().item() == 0.0
    for n, p in vs.named():
        got = p._danhip_grad.detach().float().cpu()
        assert got.shape == g0[n].shape
        assert (got - g0[n]).norm().item() <= 0.01 * g0[n].norm().item() + 1e-6, n
