"""Parity of the HBM-bound layer kernels and the loss kernels with the oracle."""
import numpy as np
import pytest
import torch

from oracle import tf_ops as T
from oracle import train as OT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", [(2, 16, 16, 64), (1, 7, 9, 128), (1, 5, 5, 8)])
def test_maxpool(shape, dev):
    from dan_amd import ops
    g = torch.Generator().manual_seed(0)
    x = torch.randn(shape, generator=g).to(torch.bfloat16)
    x[0, :2, :2] = 0.0                                           # an all-equal window: first index wins
    xr = x.float().requires_grad_(True)
    ref = T.max_pool_2x2_same(xr)
    xd = x.to(dev).requires_grad_(True)
    y = ops.max_pool_2x2(xd)
    assert torch.equal(y.float().cpu(), ref.detach())
    dy = torch.randn(ref.shape, generator=g).to(torch.bfloat16)
    y.backward(dy.to(dev))
    # oracle backward with "first maximal element in window order" tie rule
    N, H, W, C = shape
    want = torch.zeros(shape)
    xp = torch.full((N, H + H % 2, W + W % 2, C), float("-inf"))
    xp[:, :H, :W] = x.float()
    win = torch.stack([xp[:, 0::2, 0::2], xp[:, 0::2, 1::2], xp[:, 1::2, 0::2], xp[:, 1::2, 1::2]], 0)
    am = win.argmax(0)                                           # torch.argmax returns the first max
    wp = torch.zeros_like(xp)
    for t, (dh, dw) in enumerate([(0, 0), (0, 1), (1, 0), (1, 1)]):
        wp[:, dh::2, dw::2] = torch.where(am == t, dy.float(), torch.zeros(()))
    want = wp[:, :H, :W]
    assert torch.equal(xd.grad.float().cpu(), want)


@pytest.mark.parametrize("C", [256, 512])
def test_l2norm(C, dev):
    from dan_amd import ops
    g = torch.Generator().manual_seed(1)
    x = torch.relu(torch.randn((2, 6, 5, C), generator=g)).to(torch.bfloat16)
    x[0, 0, 0] = 0                                               # the 1e-10 clamp branch
    gamma = 10.0 + torch.randn((C,), generator=g)
    xr = x.float().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    ref = T.l2_normalize(xr, gr)
    xd = x.to(dev).requires_grad_(True)
    gd = gamma.to(dev).requires_grad_(True)
    y = ops.l2_normalize(xd, gd)
    assert (y.float().cpu() - ref.detach()).abs().max().item() <= 2 ** -7 * ref.abs().max().item()
    dy = torch.randn(ref.shape, generator=g).to(torch.bfloat16)
    ref.backward(dy.float())
    y.backward(dy.to(dev))
    assert (xd.grad.float().cpu() - xr.grad).abs().max().item() <= 2 ** -6 * xr.grad.abs().max().item() + 1e-3
    assert (gd.grad.cpu() - gr.grad).abs().max().item() <= 2e-3 * gr.grad.abs().max().item() + 1e-4


def test_preprocess(dev):
    from dan_amd import ops
    from oracle import nets as ON
    img = torch.randint(0, 256, (2, 9, 11, 3), dtype=torch.uint8)
    ref = ON.preprocess_synthetic(img).to(torch.bfloat16)
    got = ops.preprocess_u8(img.to(dev)).cpu()
    assert torch.equal(got[..., :3], ref) and got[..., 3:].abs().max().item() == 0


@pytest.mark.parametrize("at_least_one", [False, True])
def test_hard_negative_mining_and_losses(at_least_one, dev):
    from dan_amd import ops
    B, A = 3, 5000
    g = torch.Generator().manual_seed(4)
    cls = torch.randn((B, A, 2), generator=g) * 2
    loc = torch.randn((B, A, 4), generator=g)
    loc_t = torch.randn((B, A, 4), generator=g)
    labels = torch.zeros((B, A), dtype=torch.int64)
    labels[0, :40] = 1; labels[0, 40:80] = -1
    labels[1, 100:103] = 1
    # image 2: no positives (k = 0 unless at_least_one)
    cls[0, 200:260] = cls[0, 200:201]                            # exact ties around the threshold
    clsd = cls.to(dev).requires_grad_(True)
    locd = loc.to(dev).requires_grad_(True)
    acc = ops.detection_loss(clsd, locd, labels.to(dev).int(), loc_t.to(dev), ratio=3.0, at_least_one=at_least_one)
    acc.backward(torch.ones_like(acc))
    fn = ops._DetectionLoss
    # oracle
    clsr = cls.clone().requires_grad_(True)
    locr = loc.clone().requires_grad_(True)
    final, pos, score, k = OT.hard_neg_mask(clsr.detach(), labels, 3.0, at_least_one)
    ce, ll, _ = OT.detection_loss(clsr, locr, labels, loc_t, 3.0, at_least_one)
    (ce + ll).backward()
    ce_sum, n_sel, loc_sum, n_pos = acc.cpu().tolist()
    assert int(n_sel) == int(final.sum()) and int(n_pos) == int(pos.sum())
    assert abs(4.0 * ce_sum / n_sel - ce.item()) <= 1e-4 * abs(ce.item())
    assert abs(loc_sum / n_pos - ll.item()) <= 1e-4 * abs(ll.item())
    assert torch.allclose(clsd.grad.cpu(), clsr.grad, rtol=1e-4, atol=1e-7)
    assert torch.allclose(locd.grad.cpu(), locr.grad, rtol=1e-4, atol=1e-7)


def test_head_split_maxout(dev):
    from dan_amd import ops
    g = torch.Generator().manual_seed(5)
    B, H, W = 2, 5, 4
    for nneg, npos in ((3, 1), (1, 1), (1, 3)):
        Ch = 4 + nneg + npos
        h = torch.randn((B, H, W, Ch), generator=g)
        h[0, 0, 0, 4:4 + nneg] = 1.5                              # tie inside the max-out group
        hr = h.clone().requires_grad_(True)
        cls_ref = T.maxout_cls(hr[..., 4:], 1, nneg, npos) if nneg + npos > 2 else hr[..., 4:]
        loc_ref = hr[..., :4]
        A = H * W + 7
        hd = h.to(dev).requires_grad_(True)
        loc = torch.zeros((B, A, 4), device=dev)
        cls = torch.zeros((B, A, 2), device=dev)
        loc, cls = ops.head_split(hd, loc, cls, nneg, npos, 3)
        assert torch.equal(loc[:, 3:3 + H * W].cpu(), loc_ref.detach().reshape(B, -1, 4))
        assert torch.equal(cls[:, 3:3 + H * W].cpu(), cls_ref.detach().reshape(B, -1, 2))
        dl = torch.randn((B, A, 4), generator=g)
        dc = torch.randn((B, A, 2), generator=g)
        (loc * dl.to(dev)).sum().backward(retain_graph=True)
        (cls * dc.to(dev)).sum().backward()
        ((loc_ref.reshape(B, -1, 4) * dl[:, 3:3 + H * W]).sum() + (cls_ref.reshape(B, -1, 2) * dc[:, 3:3 + H * W]).sum()).backward()
        assert torch.allclose(hd.grad.cpu(), hr.grad, rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("N,H,W,C", [(2, 16, 16, 64), (1, 33, 47, 128), (3, 7, 9, 8), (2, 64, 64, 256)])
def test_maxpool_backward_through_arg_max_codes(N, H, W, C, dev):
    """Round 4: the 2 x 2 max-pool's backward scatters through 2-bit arg-max codes written by the forward pass instead of re-reading the
    activation (danhip_maxpool2x2_fwd_arg / _bwd_arg).  Bit-identical to the round-3 backward (danhip_maxpool2x2_bwd, first maximum in
    row-major window order = TF's MaxPoolGrad) on inputs with many exact ties (ReLU zeros, quantised values), odd sizes, accumulate."""
    from dan_amd import _lib, ops
    g = torch.Generator().manual_seed(N * 100 + H)
    x = torch.relu(torch.round(torch.randn((N, H, W, C), generator=g) * 4) / 4).to(ops.ACT).to(dev)      # ties and zeros
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    dy = torch.randn((N, Ho, Wo, C), generator=g).to(ops.ACT).to(dev)
    y0 = torch.empty((N, Ho, Wo, C), dtype=ops.ACT, device=dev)
    y1 = torch.empty_like(y0)
    arg = torch.empty((N * Ho * Wo, C // 4), dtype=torch.uint8, device=dev)
    _lib.call("danhip_maxpool2x2_fwd", _lib.ptr(x), _lib.ptr(y0), N, H, W, C, _lib.stream())
    _lib.call("danhip_maxpool2x2_fwd_arg", _lib.ptr(x), _lib.ptr(y1), _lib.ptr(arg), N, H, W, C, _lib.stream())
    assert torch.equal(y0, y1)
    for acc in (0, 1):
        old = torch.randn((N, H, W, C), generator=g).to(ops.ACT).to(dev)
        d0, d1 = old.clone(), old.clone()
        _lib.call("danhip_maxpool2x2_bwd", _lib.ptr(x), _lib.ptr(dy), _lib.ptr(d0), N, H, W, C, acc, _lib.stream())
        _lib.call("danhip_maxpool2x2_bwd_arg", _lib.ptr(arg), _lib.ptr(dy), _lib.ptr(d1), N, H, W, C, acc, _lib.stream())
        torch.cuda.synchronize()
        assert torch.equal(d0, d1), acc


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 64, 96, 64, 64), (2, 128, 128, 128, 128), (2, 66, 62, 128, 256), (1, 40, 40, 512, 512), (4, 160, 160, 256, 256)])
def test_pooling_conv_epilogues_write_the_same_arg_max_codes_as_the_pool_kernel(N, H, W, Cin, Cout, dev):
    """The fused conv + pool kernels (64 -> 64 kernel, the halo kernel's lean epilogue with and without the ReLU bit masks) derive the codes
    from the packed outputs in their epilogue registers: they must equal the pool kernel's codes of the activation the conv wrote."""
    import ctypes
    from dan_amd import _lib, ops
    g = torch.Generator().manual_seed(Cin + H)
    x = torch.randn((N, H, W, Cin), generator=g).to(ops.ACT).to(dev)
    w = (torch.randn((3, 3, Cin, Cout), generator=g) / (9 * Cin) ** 0.5).to(dev)
    b = torch.randn((Cout,), generator=g).to(dev)
    d = ops._desc(N, H, W, Cin, Cout, 3, 3, 1)
    wf, _ = ops.pack_conv_weight(d, w, need_bwd=False)
    Ho, Wo = (H + 1) // 2, (W + 1) // 2
    y = torch.empty((N, H, W, Cout), dtype=ops.ACT, device=dev)
    p = torch.empty((N, Ho, Wo, Cout), dtype=ops.ACT, device=dev)
    arg = torch.full((N * Ho * Wo, Cout // 4), 255, dtype=torch.uint8, device=dev)
    _lib.call("danhip_conv2d_fwd_pool_arg", ctypes.byref(d), _lib.ptr(x), _lib.ptr(wf), _lib.ptr(b), _lib.ptr(y), _lib.ptr(p), _lib.ptr(arg), _lib.stream())
    p_ref = torch.empty_like(p)
    arg_ref = torch.empty_like(arg)
    _lib.call("danhip_maxpool2x2_fwd_arg", _lib.ptr(y), _lib.ptr(p_ref), _lib.ptr(arg_ref), N, H, W, Cout, _lib.stream())
    torch.cuda.synchronize()
    assert torch.equal(p, p_ref) and torch.equal(arg, arg_ref)
    if _lib.lib().danhip_conv2d_fwd_emits_bits(ctypes.byref(d), 1):
        ybits = torch.empty((N * H * W, Cout // 8), dtype=torch.uint8, device=dev)
        pbits = torch.empty((N * Ho * Wo, Cout // 8), dtype=torch.uint8, device=dev)
        arg2 = torch.full_like(arg, 255)
        _lib.call("danhip_conv2d_fwd_relu_bits_arg", ctypes.byref(d), _lib.ptr(x), _lib.ptr(wf), _lib.ptr(b), _lib.ptr(y), _lib.ptr(ybits), _lib.ptr(p), _lib.ptr(pbits),
                  _lib.ptr(arg2), _lib.stream())
        torch.cuda.synchronize()
        assert torch.equal(arg2, arg_ref)


@pytest.mark.parametrize("A", [87360, 36 * 1024 + 1, 88 * 1024 + 5])
def test_hard_negative_threshold_at_1024_sized_anchor_counts_equals_topk(A, dev):
    """The k-th-largest selection of hard-negative mining at the anchor count of a 1024 x 1024 input (87 360: the 88-register instantiation of
    kth_largest_rows_kernel: 128 VGPRs, no scratch — ADVICE r5), just past the 36-register one, and past 88 * 1024 (the streaming form): the
    threshold equals torch.topk's k-th value bit for bit and the selection has exactly k negatives (train_dan.py:286-324: k = max(min(3 n_pos, n_neg), 1))."""
    import ctypes
    from dan_amd import _lib
    B = 3
    g = torch.Generator().manual_seed(A)
    cls = (torch.randn((B, A, 2), generator=g) * 3).to(dev)
    labels = torch.zeros((B, A), dtype=torch.int32)
    labels[0, :500] = 1; labels[0, 500:900] = -1
    labels[1, 7] = 1
    labels[2, ::3] = 1                                     # 3 n_pos > n_neg: k = n_neg
    labels = labels.to(dev)
    score = torch.empty((B, A), dtype=torch.float32, device=dev)
    counts = torch.empty((B, 2), dtype=torch.int32, device=dev)
    thr = torch.empty((B,), dtype=torch.float32, device=dev)
    k = torch.empty((B,), dtype=torch.int32, device=dev)
    _lib.call("danhip_hard_neg_select", _lib.ptr(cls), _lib.ptr(labels), _lib.ptr(score), _lib.ptr(counts), _lib.ptr(thr), _lib.ptr(k), B, A, 3.0, 1, _lib.stream())
    torch.cuda.synchronize()
    for b in range(B):
        n_pos, n_neg = int((labels[b] > 0).sum()), int((labels[b] == 0).sum())
        kk = max(min(3 * n_pos, n_neg), 1)
        assert counts[b].tolist() == [n_pos, n_neg] and int(k[b]) == kk
        want = torch.topk(score[b], kk).values[-1]
        assert thr[b].item() == want.item(), (A, b, kk, thr[b].item(), want.item())
        assert int(((labels[b] == 0) & (score[b] >= thr[b])).sum()) >= kk
