"""Parity of the HIP implicit-GEMM convolution (fwd / data-grad / weight-grad, through the C ABI) against the CPU
oracle's TF-SAME convolution.  bf16 storage + fp32 accumulate: inputs are rounded to bf16 first, the oracle then
computes in fp32 on exactly those values, so the only differences are accumulation order and the final bf16
rounding of the output: tolerance 2^-7 relative to the output scale (+ small abs)."""
import ctypes

import pytest
import torch

from oracle import tf_ops as T

pytestmark = pytest.mark.gpu

# (N, H, W, Cin, Cout, kh, kw, stride) — all 12 distinct backbone 3x3 shapes at reduced spatial size, 1x1, ragged edges
SHAPES = [
    (2, 24, 24, 64, 64, 3, 3, 1), (1, 20, 28, 64, 128, 3, 3, 1), (1, 16, 16, 128, 128, 3, 3, 1), (1, 16, 16, 128, 256, 3, 3, 1),
    (2, 12, 12, 256, 256, 3, 3, 1), (1, 10, 10, 256, 512, 3, 3, 1), (1, 10, 10, 512, 512, 3, 3, 1), (1, 6, 6, 512, 1024, 3, 3, 1),
    (1, 6, 6, 1024, 1024, 1, 1, 1), (1, 6, 6, 1024, 256, 1, 1, 1), (2, 20, 20, 256, 512, 3, 3, 2), (2, 10, 10, 128, 256, 3, 3, 2),
    (1, 5, 5, 128, 256, 3, 3, 2), (1, 9, 7, 64, 64, 3, 3, 1), (1, 33, 17, 8, 64, 3, 3, 1), (1, 8, 8, 256, 8, 3, 3, 1),
    (1, 8, 8, 512, 6, 3, 3, 1), (1, 12, 12, 64, 32, 3, 1, 1), (1, 12, 12, 64, 32, 1, 3, 1), (1, 7, 9, 72, 24, 3, 3, 1),
    # shapes routed to the halo-reuse 3x3 kernel: full tiles, ragged edges, 16x16 tiles, several chunks, and enough
    # items (> 256) that every persistent workgroup walks more than one (XCD-grouped mapping, NB = 2)
    (1, 16, 32, 64, 128, 3, 3, 1), (2, 30, 62, 128, 64, 3, 3, 1), (1, 32, 48, 256, 256, 3, 3, 1), (8, 64, 128, 64, 256, 3, 3, 1),
    (5, 56, 96, 128, 128, 3, 3, 1),
    # thin detection heads (Cout 8 / 6) on maps large enough for the halo kernels (loc+cls in one pass)
    (2, 16, 32, 256, 8, 3, 3, 1), (1, 32, 32, 64, 6, 3, 3, 1), (3, 24, 64, 128, 16, 3, 3, 1),
    # 64 -> 64 channels: the register-resident-weights kernel (full tiles, ragged edges, > 256 tiles)
    (2, 16, 64, 64, 64, 3, 3, 1), (1, 30, 62, 64, 64, 3, 3, 1), (9, 64, 128, 64, 64, 3, 3, 1),
    # first layer (3 channels padded to 8 -> 64): the store-bound direct kernel; ragged width, one-pixel-wide, many units
    (2, 17, 45, 8, 64, 3, 3, 1), (1, 5, 1, 8, 64, 3, 3, 1), (3, 64, 100, 8, 64, 3, 3, 1),
    # pointwise convolutions: conv_pointwise.hip (K = 64 .. 2304 as in the deformable conv) and, for ragged Cout % 64, the flat-M kernel
    (2, 48, 48, 256, 256, 1, 1, 1), (1, 64, 72, 2304, 256, 1, 1, 1), (1, 65, 67, 64, 72, 1, 1, 1), (4, 40, 40, 1024, 1024, 1, 1, 1),
    # windows on maps too small for the halo tiles, stride 2, 3x1 / 1x3, ragged pixel counts: the streaming kernel's tap-gather form
    (4, 40, 40, 256, 512, 3, 3, 1), (8, 20, 20, 512, 1024, 3, 3, 1), (2, 40, 44, 128, 64, 3, 1, 1), (2, 40, 44, 64, 128, 1, 3, 1),
    (6, 40, 40, 256, 256, 3, 3, 2), (3, 37, 41, 192, 320, 3, 3, 1), (7, 19, 23, 64, 64, 3, 3, 1),
    # channel counts that are multiples of 8 but not of 64 on the halo / row-streaming kernels (the deformable block's 256 -> 72 offsets
    # conv, net/danet_deform.py:267-290): ragged Cout (zero weight rows, guarded stores), ragged Cin (zero-filled last chunk), both
    (2, 32, 64, 256, 72, 3, 3, 1), (1, 24, 32, 128, 40, 3, 3, 1), (2, 16, 32, 72, 128, 3, 3, 1), (1, 32, 32, 136, 200, 3, 3, 1),
    (9, 40, 64, 64, 72, 3, 3, 1),
]


def _mk(shape, seed):
    N, H, W, Cin, Cout, kh, kw, s = shape
    g = torch.Generator().manual_seed(seed)
    x = (torch.randn((N, H, W, Cin), generator=g)).to(torch.bfloat16)
    w = (torch.randn((kh, kw, Cin, Cout), generator=g) / (kh * kw * Cin) ** 0.5).to(torch.bfloat16).to(torch.float32)
    b = torch.randn((Cout,), generator=g)
    return x, w, b, s


def _tol(ref):
    return 2.0 ** -7 * ref.abs().max().item() + 1e-3


@pytest.fixture(params=[True, False], ids=["splitk", "singlepass"])
def splitk(request):
    """Small maps run twice: with the scratch buffer (split-K flat-M kernel + finish pass) and without it (the halo / streaming / flat-M
    kernel of the shape in one pass, as every larger map runs)."""
    from dan_amd import ops
    old, ops.USE_SPLITK = ops.USE_SPLITK, request.param
    yield request.param
    ops.USE_SPLITK = old


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("relu", [False, True])
def test_conv_forward(shape, relu, dev, splitk):
    from dan_amd import ops
    x, w, b, s = _mk(shape, 1)
    ref = T.conv2d_same(x.float(), w, b, stride=s, relu=relu)
    y = ops.conv2d(x.to(dev), w.to(dev), b.to(dev), stride=s, relu=relu)
    torch.cuda.synchronize()
    assert y.shape == ref.shape
    err = (y.float().cpu() - ref).abs().max().item()
    assert err <= _tol(ref), (shape, err, _tol(ref))
    y32 = ops.conv2d(x.to(dev), w.to(dev), b.to(dev), stride=s, relu=relu, out_f32=True)
    err32 = (y32.cpu() - ref).abs().max().item()
    assert err32 <= 2e-3 * max(1.0, ref.abs().max().item()), (shape, err32)


@pytest.mark.parametrize("shape", SHAPES)
def test_conv_backward(shape, dev, splitk):
    from dan_amd import ops
    x, w, b, s = _mk(shape, 2)
    Cout = shape[4]
    xr = x.float().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    ref = T.conv2d_same(xr, wr, br, stride=s, relu=False)
    g = torch.Generator().manual_seed(3)
    dy = torch.randn(ref.shape, generator=g).to(torch.bfloat16)
    ref.backward(dy.float())
    xd = x.to(dev).requires_grad_(True)
    wd = w.to(dev).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True)
    y = ops.conv2d(xd, wd, bd, stride=s, relu=False)
    if Cout % 8 == 0:
        y.backward(dy.to(dev))
    else:
        y.float().backward(dy.to(dev).float())
    torch.cuda.synchronize()
    for name, got, want in (("dx", xd.grad.float().cpu(), xr.grad), ("dw", wd.grad.cpu(), wr.grad), ("db", bd.grad.cpu(), br.grad)):
        scale = want.abs().max().item() + 1e-6
        err = (got - want).abs().max().item()
        assert err <= 2.0 ** -6 * scale + 2e-3, (shape, name, err, scale)


def test_relu_mask_and_bias_grad(dev):
    """ReLU backward uses the stored bf16 output as the mask; bias gradient = column sums of the masked gradient."""
    from dan_amd import ops
    x, w, b, s = _mk((2, 12, 12, 64, 64, 3, 3, 1), 5)
    xd = x.to(dev).requires_grad_(True)
    wd = w.to(dev).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True)
    y = ops.conv2d(xd, wd, bd, stride=1, relu=True)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(7)).to(torch.bfloat16)
    y.backward(dy.to(dev))
    mask = (y.detach().float().cpu() > 0).float()
    dym = dy.float() * mask
    xr = x.float().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    T.conv2d_same(xr, wr, br, stride=1, relu=False).backward(dym)
    for name, got, want in (("dx", xd.grad.float().cpu(), xr.grad), ("dw", wd.grad.cpu(), wr.grad), ("db", bd.grad.cpu(), br.grad)):
        scale = want.abs().max().item() + 1e-6
        err = (got - want).abs().max().item()
        assert err <= 2.0 ** -6 * scale + 2e-3, (name, err, scale)


@pytest.mark.parametrize("shape", [(2, 40, 48, 256, 85, 1, 1, 1), (1, 33, 20, 64, 171, 1, 1, 1), (1, 12, 12, 64, 30, 3, 3, 1)])
def test_ragged_cout_relu_backward(shape, dev):
    """Cout not a multiple of 8 with ReLU (DAN stage-2 1x1 convs, danet.py:944-947): the ReLU mask must be taken on the
    unpadded [.., Cout] layout of y before the gradient is padded to a multiple of 8 channels."""
    from dan_amd import ops
    x, w, b, s = _mk(shape, 11)
    xd = x.to(dev).requires_grad_(True)
    wd = w.to(dev).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True)
    y = ops.conv2d(xd, wd, bd, stride=s, relu=True)
    dy = torch.randn(y.shape, generator=torch.Generator().manual_seed(7)).to(torch.bfloat16)
    y.float().backward(dy.to(dev).float())
    torch.cuda.synchronize()
    mask = (y.detach().float().cpu() > 0).float()
    xr = x.float().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    T.conv2d_same(xr, wr, br, stride=s, relu=False).backward(dy.float() * mask)
    for name, got, want in (("dx", xd.grad.float().cpu(), xr.grad), ("dw", wd.grad.cpu(), wr.grad), ("db", bd.grad.cpu(), br.grad)):
        scale = want.abs().max().item() + 1e-6
        err = (got - want).abs().max().item()
        assert err <= 2.0 ** -6 * scale + 2e-3, (shape, name, err, scale)


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(16, 1024, 1024, 64, 64), (8, 1024, 1024, 128, 128)])
def test_two_gib_activations_batch_equals_its_halves(N, H, W, Cin, Cout, dev):
    """BASELINE.json configs[3..4] run at 1024x1024: conv1_2 / conv2_2 activations reach 2 GiB (byte offsets beyond 2^31).
    Size-independent property: a batch-N launch equals its two batch-N/2 launches image for image (fwd, dgrad bit-exact;
    the weight gradient is their sum up to fp32 atomics order)."""
    import ctypes
    from dan_amd import ops
    from dan_amd._lib import BF16, call, ptr, stream
    g = torch.Generator(device=dev).manual_seed(1)
    x = torch.randn((N, H, W, Cin), generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    dy = torch.randn((N, H, W, Cout), generator=g, device=dev, dtype=torch.float32).to(torch.bfloat16)
    w = torch.randn((3, 3, Cin, Cout), generator=g, device=dev) / (9 * Cin) ** 0.5
    b = torch.randn((Cout,), generator=g, device=dev)
    assert x.numel() * 2 >= 2 ** 31

    def all3(xx, dd):
        n = xx.shape[0]
        d = ops._desc(n, H, W, Cin, Cout, 3, 3, 1)
        wf, wb = ops.pack_conv_weight(d, w, need_bwd=True)
        y = torch.empty((n, H, W, Cout), dtype=torch.bfloat16, device=dev)
        dx = torch.empty_like(xx)
        dw = torch.zeros((3, 3, Cin, Cout), dtype=torch.float32, device=dev)
        call("danhip_conv2d_fwd", ctypes.byref(d), ptr(xx), ptr(wf), ptr(b), ptr(y), BF16, 1, None, stream())
        call("danhip_conv2d_bwd_data", ctypes.byref(d), ptr(dd), ptr(wb), ptr(xx), ptr(dx), 0, stream())
        call("danhip_conv2d_bwd_weight", ctypes.byref(d), ptr(xx), ptr(dd), ptr(dw), None, Cin, stream())
        return y, dx, dw

    y, dx, dw = all3(x, dy)
    h = N // 2
    ya, dxa, dwa = all3(x[:h].contiguous(), dy[:h].contiguous())
    yb, dxb, dwb = all3(x[h:].contiguous(), dy[h:].contiguous())
    assert torch.equal(y[:h], ya) and torch.equal(y[h:], yb)
    assert torch.equal(dx[:h], dxa) and torch.equal(dx[h:], dxb)
    assert ((dw - dwa - dwb).abs().max() / dw.abs().max()).item() < 1e-5


@pytest.mark.parametrize("shape", [(2, 16, 64, 64, 64), (1, 30, 62, 64, 64), (3, 32, 64, 64, 128), (2, 24, 64, 128, 128), (1, 31, 45, 128, 256),
                                   (2, 32, 32, 256, 512), (1, 37, 29, 256, 512), (2, 10, 10, 512, 512), (1, 12, 12, 64, 32)])
def test_conv_relu_with_fused_max_pool(shape, dev):
    """danhip_conv2d_fwd_pool: the pooled map written by the conv epilogue (64->64 kernel, 8x32 and 16x16 halo tiles; other shapes
    fall back to the pool kernel inside the call) is bit-identical to max_pool_2x2 of the conv output, ragged edges included,
    and the gradient path is the unfused one."""
    from dan_amd import ops
    N, H, W, Cin, Cout = shape
    g = torch.Generator().manual_seed(H * W + Cout)
    x = torch.randn((N, H, W, Cin), generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn((3, 3, Cin, Cout), generator=g) / (9 * Cin) ** 0.5).to(dev)
    b = torch.randn((Cout,), generator=g).to(dev)
    outs = []
    for fused in (False, True):
        xd = x.clone().requires_grad_(True)
        wd = w.clone().requires_grad_(True)
        bd = b.clone().requires_grad_(True)
        y = ops.conv2d(xd, wd, bd, relu=True, pool=fused)
        assert hasattr(y, "_dh_pooled") == fused
        p = ops.max_pool_2x2(y)
        gen = torch.Generator().manual_seed(7)
        dp = torch.randn(p.shape, generator=gen).to(torch.bfloat16).to(dev)
        p.backward(dp)
        outs.append((y.detach(), p.detach(), xd.grad, wd.grad))
    (y0, p0, dx0, dw0), (y1, p1, dx1, dw1) = outs
    assert p1.shape == (N, (H + 1) // 2, (W + 1) // 2, Cout)
    assert torch.equal(y0, y1) and torch.equal(p0, p1)
    assert torch.equal(dx0, dx1)
    assert torch.allclose(dw0, dw1, rtol=1e-3, atol=1e-3 * dw0.abs().max().item())


@pytest.mark.parametrize("accumulate", [0, 1])
def test_pointwise_data_gradient_ragged_channels(accumulate, dev):
    """danhip_conv2d_bwd_data of a 1x1 conv whose channel counts are not multiples of 64 (flat-M kernel), with / without the fused ReLU
    mask and accumulation — against the fp32 formula dx = dy . W^T."""
    import ctypes
    from dan_amd import ops
    from dan_amd._lib import call, ptr, stream
    N, H, W, Cin, Cout = 2, 48, 50, 320, 136
    g = torch.Generator().manual_seed(5)
    w = (torch.randn((1, 1, Cin, Cout), generator=g) / Cin ** 0.5).to(torch.bfloat16).float()
    dy = torch.randn((N, H, W, Cout), generator=g).to(torch.bfloat16)
    x = torch.randn((N, H, W, Cin), generator=g).to(torch.bfloat16)
    dx0 = torch.randn((N, H, W, Cin), generator=g).to(torch.bfloat16)
    d = ops._desc(N, H, W, Cin, Cout, 1, 1, 1)
    _, wb = ops.pack_conv_weight(d, w.to(dev), need_bwd=True)
    ref = dy.float().reshape(-1, Cout) @ w.reshape(Cin, Cout).t()
    ref = ref.reshape(N, H, W, Cin) + (dx0.float() if accumulate else 0.0)
    for mask in (None, x.to(dev)):
        dx = dx0.to(dev).clone()
        call("danhip_conv2d_bwd_data", ctypes.byref(d), ptr(dy.to(dev)), ptr(wb), ptr(mask), ptr(dx), accumulate, stream())
        torch.cuda.synchronize()
        want = ref if mask is None else torch.where(x.float() > 0, ref, dx0.float() if accumulate else torch.zeros_like(ref))
        err = (dx.float().cpu() - want).abs().max().item()
        assert err <= 2.0 ** -7 * want.abs().max().item() + 2e-2, (mask is not None, err)


# The 12 distinct 3x3 shapes of the VGG-16 backbone at BASELINE.json's 640x640 (SURVEY §7 "minimum slice", Appendix B) plus conv1_1 and
# the two heavy pointwise shapes, at N = 1 and FULL spatial size, against the CPU oracle convolution (oneDNN fp32 on bf16-rounded
# operands): forward with bias + ReLU, data gradient, weight gradient, bias gradient.
def test_ragged_channel_counts_run_on_the_halo_and_row_streaming_kernels(dev, splitk):
    import ctypes
    from dan_amd import ops
    from dan_amd._lib import lib
    d = ops._desc(2, 32, 64, 256, 72, 3, 3, 1)                  # (| 16: the single-pass call; maps this small split K when given scratch)
    assert lib().danhip_conv_kernel_label(ctypes.byref(d), 0 | 16).decode().startswith("conv3x3_halo_kernel<8, 32, 128")
    assert lib().danhip_conv_kernel_label(ctypes.byref(d), 1 | 16).decode().startswith("conv3x3_halo_kernel<8, 32, 128")
    assert lib().danhip_conv_wgrad_kernel_label(ctypes.byref(d)).decode() == "conv_wgrad_rows_kernel<128>"
    d = ops._desc(1, 24, 32, 128, 40, 3, 3, 1)
    assert lib().danhip_conv_kernel_label(ctypes.byref(d), 0 | 16).decode().startswith("conv3x3_halo_kernel<8, 32, 64")
    assert lib().danhip_conv_wgrad_kernel_label(ctypes.byref(d)).decode() == "conv_wgrad_rows_kernel<64>"


FULL_SHAPES = [
    ("conv1_1", 640, 8, 64, 3, 1), ("conv1_2", 640, 64, 64, 3, 1), ("conv2_1", 320, 64, 128, 3, 1), ("conv2_2", 320, 128, 128, 3, 1),
    ("conv3_1", 160, 128, 256, 3, 1), ("conv3_2", 160, 256, 256, 3, 1), ("conv4_1", 80, 256, 512, 3, 1), ("conv4_2", 80, 512, 512, 3, 1),
    ("conv5_x", 40, 512, 512, 3, 1), ("fc6", 20, 512, 1024, 3, 1), ("conv6_2", 20, 256, 512, 3, 2), ("conv7_2", 10, 128, 256, 3, 2),
    ("fc7", 20, 1024, 1024, 1, 1), ("lfpn_lateral", 160, 256, 256, 1, 1),
]


@pytest.mark.parametrize("name,hw,cin,cout,k,stride", FULL_SHAPES)
def test_backbone_layer_at_full_size_vs_oracle(name, hw, cin, cout, k, stride, dev):
    from dan_amd import ops
    x, w, b, s = _mk((1, hw, hw, cin, cout, k, k, stride), 100 + hw + cin)
    if name == "conv1_1":                                         # 3 real channels padded to 8 (prepare_input): the pad lanes are zero
        x[..., 3:] = 0
    xr = x.float().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    pre = T.conv2d_same(xr, wr, br, stride=s, relu=False)
    ref = torch.relu(pre)
    g = torch.Generator().manual_seed(3)
    dy = torch.randn(ref.shape, generator=g).to(torch.bfloat16)
    xd = x.to(dev).requires_grad_(True)
    wd = w.to(dev).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True)
    y = ops.conv2d(xd, wd, bd, stride=s, relu=True)
    y.backward(dy.to(dev))
    torch.cuda.synchronize()
    err = (y.detach().float().cpu() - ref.detach()).abs().max().item()
    assert err <= _tol(ref.detach()), (name, "fwd", err, _tol(ref.detach()))
    # the HIP path masks by its own stored bf16 output; use the same mask so that sign ties at the ReLU boundary do not enter
    mask = (y.detach().float().cpu() > 0).float()
    pre.backward(dy.float() * mask)
    checks = [("dw", wd.grad.cpu(), wr.grad), ("db", bd.grad.cpu(), br.grad)]
    if name != "conv1_1":                                         # the image needs no gradient (and the pad channels have none)
        checks.append(("dx", xd.grad.float().cpu(), xr.grad))
    for what, got, want in checks:
        scale = want.abs().max().item() + 1e-6
        e = (got - want).abs().max().item()
        assert e <= 2.0 ** -6 * scale + 2e-3, (name, what, e, scale)


# conv_pointwise.hip (1x1 / stride 1, channels multiples of 64): tiles 256 / 128 / 64 wide, K = 64 .. 2304 (the deformable conv's GEMM),
# ragged pixel counts (last 128-pixel tile partial), one K-step per item (C = 64), several column blocks (Co = 512, 1024)
PW_SHAPES = [(2, 48, 48, 256, 256), (1, 64, 72, 2304, 256), (4, 40, 40, 1024, 1024), (1, 45, 47, 64, 64), (3, 33, 35, 64, 256), (1, 50, 50, 128, 512),
             (2, 40, 56, 512, 128), (1, 61, 67, 192, 64), (5, 32, 32, 256, 64), (1, 160, 160, 256, 256), (2, 47, 49, 320, 192), (1, 70, 70, 128, 72)]


@pytest.mark.parametrize("shape", PW_SHAPES)
def test_pointwise_kernel_forward_and_data_gradient(shape, dev, splitk):
    """Forward (bias, ReLU on / off) against the oracle convolution; data gradient through the C ABI in all four epilogue modes
    (plain, ReLU mask of the producer, accumulate into an existing gradient, both) against dx = dy . W^T in fp32."""
    import ctypes
    from dan_amd import ops
    from dan_amd._lib import call, lib, ptr, stream
    N, H, W, Cin, Cout = shape
    g = torch.Generator().manual_seed(N * H + Cin + Cout)
    x = torch.randn((N, H, W, Cin), generator=g).to(torch.bfloat16)
    w = (torch.randn((1, 1, Cin, Cout), generator=g) / Cin ** 0.5).to(torch.bfloat16).float()
    b = torch.randn((Cout,), generator=g)
    d = ops._desc(N, H, W, Cin, Cout, 1, 1, 1)
    if Cin % 64 == 0 and Cout % 64 == 0:                       # (| 16: the single-pass call — small maps split K when given scratch)
        assert lib().danhip_conv_kernel_label(ctypes.byref(d), 0 | 16).decode().startswith("conv_pointwise_kernel")
        assert lib().danhip_conv_kernel_label(ctypes.byref(d), 5 | 16).decode().startswith("conv_pointwise_kernel")
    if Cin % 64 == 0 and Cin >= 128 and N * H * W >= 4096:
        assert lib().danhip_conv_wgrad_kernel_label(ctypes.byref(d)).decode() == "conv_wgrad_pw_kernel"
    for relu in (False, True):
        ref = T.conv2d_same(x.float(), w, b, stride=1, relu=relu)
        y = ops.conv2d(x.to(dev), w.to(dev), b.to(dev), stride=1, relu=relu)
        torch.cuda.synchronize()
        err = (y.float().cpu() - ref).abs().max().item()
        assert err <= _tol(ref), (shape, relu, err, _tol(ref))
    _, wb = ops.pack_conv_weight(d, w.to(dev), need_bwd=True)
    dy = torch.randn((N, H, W, Cout), generator=g).to(torch.bfloat16)
    dx0 = torch.randn((N, H, W, Cin), generator=g).to(torch.bfloat16)
    ref = (dy.float().reshape(-1, Cout) @ w.reshape(Cin, Cout).t()).reshape(N, H, W, Cin)
    xd, dyd, dx0d = x.to(dev), dy.to(dev), dx0.to(dev)       # (held: a temporary passed through ptr() would be freed before the launch)
    for masked in (False, True):
        for acc in (0, 1):
            dx = dx0d.clone()
            call("danhip_conv2d_bwd_data", ctypes.byref(d), ptr(dyd), ptr(wb), ptr(xd) if masked else None, ptr(dx), acc, stream())
            torch.cuda.synchronize()
            want = torch.where(x.float() > 0, ref, torch.zeros_like(ref)) if masked else ref
            want = want + (dx0.float() if acc else 0.0)
            err = (dx.float().cpu() - want).abs().max().item()
            assert err <= 2.0 ** -7 * want.abs().max().item() + 2e-2, (shape, masked, acc, err)
    # weight / bias gradient (conv_wgrad_pw.hip where eligible): dW = X^T dY, db = column sums of dY, accumulated onto what is there
    dw = torch.ones((1, 1, Cin, Cout), dtype=torch.float32, device=dev)
    db = torch.ones((Cout,), dtype=torch.float32, device=dev)
    call("danhip_conv2d_bwd_weight", ctypes.byref(d), ptr(xd), ptr(dyd), ptr(dw), ptr(db), Cin, stream())
    torch.cuda.synchronize()
    want_w = x.float().reshape(-1, Cin).t() @ dy.float().reshape(-1, Cout) + 1.0
    want_b = dy.float().reshape(-1, Cout).sum(0) + 1.0
    assert (dw.cpu().reshape(Cin, Cout) - want_w).abs().max().item() <= 2e-3 * want_w.abs().max().item(), shape
    assert (db.cpu() - want_b).abs().max().item() <= 2e-3 * want_b.abs().max().item() + 1e-2, shape


@pytest.mark.parametrize("kh,kw,cin,cin_real,cout", [(3, 3, 64, 64, 128), (3, 3, 8, 3, 64), (1, 1, 256, 256, 85), (3, 3, 256, 256, 72), (3, 3, 512, 512, 6),
                                                     (3, 1, 64, 64, 32), (1, 1, 1024, 1024, 1024), (3, 3, 72, 72, 200)])
def test_weight_packing_layouts(kh, kw, cin, cin_real, cout, dev):
    """danhip_pack_conv_weight (one weight) and the batched form (one launch for a table of weights) against the layouts DESIGN §1 states:
    wf[co][tap*Cin + c] = w[tap][c][co], wb[ci][tap_flipped*Cout8 + co] = w[tap][ci][co], zeros in every padding row / column."""
    import ctypes
    from dan_amd import ops, _lib
    from dan_amd._lib import call, ptr, stream
    g = torch.Generator().manual_seed(kh * 100 + cin + cout)
    w = torch.randn((kh, kw, cin_real, cout), generator=g)
    d = ops._desc(1, 8, 8, cin, cout, kh, kw, 1)
    (rf, cf), (rb, cb) = ops.packed_dims(d, 0), ops.packed_dims(d, 1)
    taps, co8 = kh * kw, (cout + 7) // 8 * 8
    wt = w.to(torch.bfloat16).reshape(taps, cin_real, cout)
    want_f = torch.zeros((rf, cf), dtype=torch.bfloat16)
    want_b = torch.zeros((rb, cb), dtype=torch.bfloat16)
    for t in range(taps):
        want_f[:cout, t * cin:t * cin + cin_real] = wt[t].t()
        fi, fj = divmod(t, kw)
        tf_ = (kh - 1 - fi) * kw + (kw - 1 - fj)                          # the data gradient runs the taps flipped
        want_b[:cin_real, tf_ * co8:tf_ * co8 + cout] = wt[t]
    wd = w.to(dev)
    wf = torch.full((rf, cf), 7.0, dtype=torch.bfloat16, device=dev)
    wb = torch.full((rb, cb), 7.0, dtype=torch.bfloat16, device=dev)
    call("danhip_pack_conv_weight", ctypes.byref(d), ptr(wd), cin_real, ptr(wf), ptr(wb), stream())
    assert torch.equal(wf.cpu(), want_f) and torch.equal(wb.cpu(), want_b)
    # batched: this weight twice in one table (second entry forward-only), non-zero first_block for the second
    wf1, wb1, wf2 = torch.full_like(wf, 3.0), torch.full_like(wb, 3.0), torch.full_like(wf, 3.0)
    arr = (_lib.PackEntry * 2)()
    nb = ctypes.c_int32()
    call("danhip_pack_entry_init", ctypes.byref(arr[0]), ctypes.byref(d), ptr(wd), cin_real, ptr(wf1), ptr(wb1), 0, ctypes.byref(nb))
    first = nb.value
    call("danhip_pack_entry_init", ctypes.byref(arr[1]), ctypes.byref(d), ptr(wd), cin_real, ptr(wf2), None, first, ctypes.byref(nb))
    tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
    call("danhip_pack_conv_weights_batched", ptr(tab), 2, first + nb.value, stream())
    torch.cuda.synchronize()
    assert torch.equal(wf1.cpu(), want_f) and torch.equal(wb1.cpu(), want_b) and torch.equal(wf2.cpu(), want_f)


@pytest.mark.parametrize("M,C", [(1000, 256), (77, 8), (513, 72), (64, 1024), (5, 40)])
def test_relu_bits_kernel(M, C, dev):
    import numpy as np
    from dan_amd._lib import call, ptr, stream
    g = torch.Generator().manual_seed(M + C)
    x = torch.randn((M, C), generator=g).to(torch.bfloat16)
    x[::3] = 0
    x[1, :] = -0.0
    bits = torch.full((M, C // 8), 0xAA, dtype=torch.uint8, device=dev)
    call("danhip_relu_bits", ptr(x.to(dev)), ptr(bits), M, C, stream())
    want = np.packbits((x.float().numpy() > 0), axis=1, bitorder="little")
    assert np.array_equal(bits.cpu().numpy(), want)


@pytest.mark.parametrize("shape", [(2, 24, 64, 128, 128), (1, 33, 62, 64, 256), (3, 16, 16, 256, 128), (1, 80, 80, 512, 256), (2, 30, 62, 72, 128), (9, 64, 96, 128, 256)])
@pytest.mark.parametrize("accumulate", [0, 1])
def test_data_gradient_with_bit_mask_equals_the_16bit_mask_form(shape, accumulate, dev):
    """danhip_conv2d_bwd_data_bits (the ReLU mask staged in LDS as one bit per element) against danhip_conv2d_bwd_data with the activation
    itself as the mask: the same kernel and arithmetic, so the gradients are identical bit for bit; several items per workgroup, ragged
    edges, 16x16 tiles, accumulate on / off, a ragged input-channel count of the forward conv's OUTPUT (dY with 72 channels)."""
    import ctypes
    from dan_amd import ops
    from dan_amd._lib import call, lib, ptr, stream
    N, H, W, Cout, Cin = shape                     # forward conv Cin -> Cout; its data gradient produces Cin channels
    g = torch.Generator().manual_seed(sum(shape))
    d = ops._desc(N, H, W, Cin, Cout, 3, 3, 1)
    assert lib().danhip_conv2d_bwd_data_takes_bits(ctypes.byref(d)) == 1
    w = (torch.randn((3, 3, Cin, Cout), generator=g) / (9 * Cin) ** 0.5).to(dev)
    _, wb = ops.pack_conv_weight(d, w, need_bwd=True)
    x = torch.relu(torch.randn((N, H, W, Cin), generator=g)).to(torch.bfloat16).to(dev)
    dy = torch.randn((N, H, W, Cout), generator=g).to(torch.bfloat16).to(dev)
    old = torch.randn((N, H, W, Cin), generator=g).to(torch.bfloat16).to(dev)
    bits = torch.empty((N * H * W, Cin // 8), dtype=torch.uint8, device=dev)
    call("danhip_relu_bits", ptr(x), ptr(bits), N * H * W, Cin, stream())
    a, b = old.clone(), old.clone()
    call("danhip_conv2d_bwd_data", ctypes.byref(d), ptr(dy), ptr(wb), ptr(x), ptr(a), accumulate, stream())
    call("danhip_conv2d_bwd_data_bits", ctypes.byref(d), ptr(dy), ptr(wb), ptr(bits), ptr(b), accumulate, stream())
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    assert (a.float().abs().sum() > 0) and ((a == 0) | (x > 0)).all() if not accumulate else True


@pytest.mark.parametrize("shape", [(2, 24, 64), (1, 30, 62), (9, 64, 128), (3, 40, 96)])
@pytest.mark.parametrize("accumulate", [0, 1])
def test_first_layer_writes_the_bit_mask_and_the_64_channel_data_gradient_reads_it(shape, accumulate, dev):
    """conv1_1 (3 -> 64, the store-bound kernel) writes the ReLU bit mask of its output beside it; conv1_2's data gradient (64 -> 64, register-
    resident weights) stages 2 KiB of those bits per tile instead of the 64 KiB 16-bit mask tile: bit masks equal danhip_relu_bits of the
    written tensor, gradients equal the 16-bit-mask form bit for bit (ragged tile edges, several tiles per workgroup, accumulate)."""
    import ctypes
    from dan_amd import ops
    from dan_amd._lib import call, lib, ptr, stream
    N, H, W = shape
    g = torch.Generator().manual_seed(sum(shape))
    d1 = ops._desc(N, H, W, 8, 64, 3, 3, 1)
    assert lib().danhip_conv2d_fwd_emits_bits(ctypes.byref(d1), 0) == 1 and lib().danhip_conv2d_fwd_emits_bits(ctypes.byref(d1), 1) == 0
    x0 = torch.zeros((N, H, W, 8), dtype=torch.bfloat16)
    x0[..., :3] = torch.randn((N, H, W, 3), generator=g).to(torch.bfloat16)
    w1 = (torch.randn((3, 3, 3, 64), generator=g) / 27 ** 0.5).to(dev)
    b1 = (0.2 * torch.randn(64, generator=g)).to(dev)
    wf, _ = ops.pack_conv_weight(d1, w1, need_bwd=False)
    y = torch.empty((N, H, W, 64), dtype=torch.bfloat16, device=dev)
    y_ref = torch.empty_like(y)
    bits = torch.full((N * H * W, 8), 0x55, dtype=torch.uint8, device=dev)
    x0d = x0.to(dev)
    call("danhip_conv2d_fwd_relu_bits", ctypes.byref(d1), ptr(x0d), ptr(wf), ptr(b1), ptr(y), ptr(bits), None, None, stream())
    call("danhip_conv2d_fwd", ctypes.byref(d1), ptr(x0d), ptr(wf), ptr(b1), ptr(y_ref), 1, 1, None, stream())
    want = torch.empty_like(bits)
    call("danhip_relu_bits", ptr(y), ptr(want), N * H * W, 64, stream())
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref) and torch.equal(bits, want)
    assert 0.2 < (y > 0).float().mean().item() < 0.8
    d2 = ops._desc(N, H, W, 64, 64, 3, 3, 1)
    assert lib().danhip_conv2d_bwd_data_takes_bits(ctypes.byref(d2)) == 1
    w2 = (torch.randn((3, 3, 64, 64), generator=g) / 576 ** 0.5).to(dev)
    _, wb = ops.pack_conv_weight(d2, w2, need_bwd=True)
    dy = torch.randn((N, H, W, 64), generator=g).to(torch.bfloat16).to(dev)
    old = torch.randn((N, H, W, 64), generator=g).to(torch.bfloat16).to(dev)
    a, b = old.clone(), old.clone()
    call("danhip_conv2d_bwd_data", ctypes.byref(d2), ptr(dy), ptr(wb), ptr(y), ptr(a), accumulate, stream())
    call("danhip_conv2d_bwd_data_bits", ctypes.byref(d2), ptr(dy), ptr(wb), ptr(bits), ptr(b), accumulate, stream())
    torch.cuda.synchronize()
    assert torch.equal(a, b)
    d3 = ops._desc(1, 30, 47, 64, 64, 3, 3, 1)      # odd width: pixel pairs would straddle rows, the kernel keeps the 16-bit mask
    assert lib().danhip_conv2d_bwd_data_takes_bits(ctypes.byref(d3)) == 0


@pytest.mark.parametrize("shape", [(2, 24, 64, 64, 128), (1, 33, 62, 128, 256), (3, 16, 16, 256, 128), (9, 64, 96, 128, 128)])
@pytest.mark.parametrize("pool", [False, True])
def test_forward_kernel_writes_the_relu_bit_masks(shape, pool, dev):
    """The 128-wide halo forward tiles leave the ReLU bit mask of their output (and of the fused pooled map) for the next convolution's
    data gradient: identical to danhip_relu_bits of the tensors they wrote, ragged edges and odd sizes included; the hand-off reaches
    the consumer (its data gradient equals the one computed with the 16-bit mask, bit for bit)."""
    import numpy as np
    from dan_amd import ops
    N, H, W, Cin, Cout = shape
    g = torch.Generator().manual_seed(sum(shape) + pool)
    x = torch.randn((N, H, W, Cin), generator=g).to(torch.bfloat16).to(dev).requires_grad_(True)
    w = (torch.randn((3, 3, Cin, Cout), generator=g) / (9 * Cin) ** 0.5).to(dev).requires_grad_(True)
    b = (0.1 * torch.randn(Cout, generator=g)).to(dev).requires_grad_(True)
    w2 = (torch.randn((3, 3, Cout, 128), generator=g) / (9 * Cout) ** 0.5).to(dev).requires_grad_(True)
    res = {}
    for use in (True, False):
        ops.USE_RELU_BITS = use
        try:
            for t in (x, w, b, w2):
                t.grad = None
            y = ops.conv2d(x, w, b, relu=True, pool=pool)
            z = ops.max_pool_2x2(y) if pool else y
            if use:
                for t in ((y, z) if pool else (y,)):
                    bits = t._dh_bits[0]
                    want = np.packbits(t.detach().float().cpu().numpy().reshape(-1, Cout) > 0, axis=1, bitorder="little")
                    assert np.array_equal(bits.cpu().numpy(), want)
            else:
                assert getattr(y, "_dh_bits", None) is None
            if z.shape[1] >= 16 and z.shape[2] >= 16:
                out = ops.conv2d(z, w2, None, relu=False)
                out.backward(torch.ones_like(out))
                res[use] = (x.grad.clone(), w.grad.clone())
        finally:
            ops.USE_RELU_BITS = True
    if res:
        assert torch.equal(res[True][0], res[False][0])
        assert (res[True][1] - res[False][1]).abs().max().item() <= 1e-4 * res[False][1].abs().max().item()      # (fp32 atomics order)


@pytest.mark.parametrize("shape", [(2, 32, 64, 64, 128, 3, 3, 1), (3, 40, 40, 256, 256, 3, 3, 1), (2, 16, 32, 256, 8, 3, 3, 1), (2, 32, 64, 256, 72, 3, 3, 1),
                                   (5, 56, 96, 128, 128, 3, 3, 1), (1, 33, 47, 64, 64, 3, 3, 1),
                                   # pointwise (conv_wgrad_pw.hip): whole tile, ragged ci / co tiles (inactive waves), K = 2304, several tiles
                                   (2, 48, 48, 256, 256, 1, 1, 1), (1, 64, 72, 2304, 256, 1, 1, 1), (2, 80, 80, 512, 64, 1, 1, 1), (3, 40, 40, 320, 192, 1, 1, 1),
                                   (2, 50, 50, 128, 512, 1, 1, 1)])
def test_weight_gradient_slab_form_equals_the_atomic_form(shape, dev):
    """danhip_conv2d_bwd_weight_ws (partial tiles as plain stores + a combine pass) against danhip_conv2d_bwd_weight (fp32 atomics): same
    += semantics on a pre-filled gradient, equal up to fp32 summation order; both against the oracle convolution's weight gradient."""
    import ctypes
    from dan_amd import ops
    from dan_amd._lib import call, lib, ptr, stream
    N, H, W, Cin, Cout, kh, kw, s = shape
    x, w, b, _ = _mk(shape, 21)
    co8 = (Cout + 7) // 8 * 8
    g = torch.Generator().manual_seed(22)
    dy = torch.zeros((N, H, W, co8), dtype=torch.bfloat16)
    dy[..., :Cout] = torch.randn((N, H, W, Cout), generator=g).to(torch.bfloat16)
    d = ops._desc(N, H, W, Cin, Cout, kh, kw, s)
    nws = lib().danhip_conv2d_bwd_weight_workspace_bytes(ctypes.byref(d))
    assert nws > 0, "the row-streaming kernel takes this shape"
    xd, dyd = x.to(dev), dy.to(dev)
    pre = torch.randn((kh, kw, Cin, Cout), generator=g).to(dev)
    dw_a, dw_s = pre.clone(), pre.clone()
    db_a, db_s = torch.zeros(Cout, device=dev), torch.zeros(Cout, device=dev)
    ws = torch.full((nws,), 0x7f, dtype=torch.uint8, device=dev)               # garbage in the scratch buffer must not matter
    call("danhip_conv2d_bwd_weight", ctypes.byref(d), ptr(xd), ptr(dyd), ptr(dw_a), ptr(db_a), Cin, stream())
    call("danhip_conv2d_bwd_weight_ws", ctypes.byref(d), ptr(xd), ptr(dyd), ptr(dw_s), ptr(db_s), Cin, ptr(ws), nws, stream())
    torch.cuda.synchronize()
    xr = x.float()
    wr = w.clone().requires_grad_(True)
    ref = T.conv2d_same(xr, wr, None, stride=s)
    ref.backward(dy[..., :Cout].float())
    want = wr.grad + pre.cpu()
    scale = wr.grad.abs().max().item()
    assert (dw_s.cpu() - want).abs().max().item() <= 2.0 ** -6 * scale + 2e-3
    assert (dw_s - dw_a).abs().max().item() <= 1e-4 * scale + 1e-5
    assert (db_s - db_a).abs().max().item() <= 1e-4 * db_a.abs().max().item() + 1e-5
    # a buffer that is too small is refused in favour of the atomic form, never overrun
    dw_t = pre.clone()
    call("danhip_conv2d_bwd_weight_ws", ctypes.byref(d), ptr(xd), ptr(dyd), ptr(dw_t), None, Cin, ptr(ws), nws // 2, stream())
    torch.cuda.synchronize()
    assert (dw_t - dw_a).abs().max().item() <= 1e-4 * scale + 1e-5


@pytest.mark.parametrize("shape", [(2, 20, 20, 512, 1024, 3, 3, 1), (2, 20, 20, 1024, 1024, 1, 1, 1), (2, 40, 40, 512, 512, 3, 3, 1), (2, 20, 20, 256, 512, 3, 3, 2),
                                   (2, 40, 40, 512, 6, 3, 3, 1), (2, 5, 5, 256, 6, 3, 3, 1), (1, 80, 80, 512, 256, 3, 3, 1), (2, 10, 10, 128, 256, 3, 3, 2),
                                   (3, 37, 41, 192, 320, 3, 3, 1), (2, 24, 40, 72, 128, 3, 3, 1)])
def test_splitk_form_equals_the_single_pass_form(shape, dev):
    """danhip_conv2d_{fwd,bwd_data}_ws with the scratch buffer (K split over workgroups + finish pass) against the same call without it, on the
    per-rank shapes of the strong-scaling series (2 images per GPU): forward with bias / ReLU (16-bit and fp32 outputs), data gradient with
    mask and accumulate.  Equal up to fp32 summation order and one 16-bit rounding."""
    import ctypes
    from dan_amd import ops
    from dan_amd._lib import BF16, F32, call, lib, ptr, stream
    N, H, W, Cin, Cout, kh, kw, s = shape
    x, w, b, _ = _mk(shape, 31)
    d = ops._desc(N, H, W, Cin, Cout, kh, kw, s)
    xd, bd = x.to(dev), b.to(dev)
    wf, wb = ops.pack_conv_weight(d, w.to(dev), need_bwd=True)
    n0, n1 = lib().danhip_conv2d_workspace_bytes(ctypes.byref(d), 0), lib().danhip_conv2d_workspace_bytes(ctypes.byref(d), 1)
    assert n0 > 0 or n1 > 0, "neither direction of this shape splits"
    ws = torch.full((max(n0, n1, 16),), 0x7f, dtype=torch.uint8, device=dev)
    co8 = (Cout + 7) // 8 * 8
    for out_f32 in ([False, True] if Cout % 8 == 0 else [True]):
        y0 = torch.empty((N, d.Ho, d.Wo, Cout), dtype=torch.float32 if out_f32 else torch.bfloat16, device=dev)
        y1 = torch.empty_like(y0)
        call("danhip_conv2d_fwd", ctypes.byref(d), ptr(xd), ptr(wf), ptr(bd), ptr(y0), F32 if out_f32 else BF16, 1, None, stream())
        call("danhip_conv2d_fwd_ws", ctypes.byref(d), ptr(xd), ptr(wf), ptr(bd), ptr(y1), F32 if out_f32 else BF16, 1, None, ptr(ws), n0, stream())
        torch.cuda.synchronize()
        tol = (1e-4 if out_f32 else 2.0 ** -7) * y0.float().abs().max().item() + 1e-5
        assert (y0.float() - y1.float()).abs().max().item() <= tol, (shape, out_f32)
    g = torch.Generator().manual_seed(32)
    dy = torch.zeros((N, d.Ho, d.Wo, co8), dtype=torch.bfloat16)
    dy[..., :Cout] = torch.randn((N, d.Ho, d.Wo, Cout), generator=g).to(torch.bfloat16)
    dyd = dy.to(dev)
    old = torch.randn((N, H, W, Cin), generator=g).to(torch.bfloat16).to(dev)
    for mask, acc in ((None, 0), (xd, 0), (xd, 1)):
        dx0, dx1 = old.clone(), old.clone()
        call("danhip_conv2d_bwd_data", ctypes.byref(d), ptr(dyd), ptr(wb), ptr(mask), ptr(dx0), acc, stream())
        call("danhip_conv2d_bwd_data_ws", ctypes.byref(d), ptr(dyd), ptr(wb), ptr(mask), ptr(dx1), acc, ptr(ws), n1, stream())
        torch.cuda.synchronize()
        tol = 2.0 ** -7 * dx0.float().abs().max().item() + 1e-5
        assert (dx0.float() - dx1.float()).abs().max().item() <= tol, (shape, mask is not None, acc)


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(4, 160, 160, 128, 256), (3, 83, 121, 192, 136)])
def test_one_barrier_per_step_is_bit_identical_to_the_two_barrier_form(N, H, W, Cin, Cout, dev):
    """Round 3 dropped the second workgroup barrier per step of the 3x3 tile kernels (conv_halo.hip main-loop comment: b1 alone carries the
    LDS hand-offs).  Forward and data gradient have no atomics, so the two forms must agree bit for bit, repeatedly, also while another
    stream keeps the chip busy (tools/stress_determinism.py is the long version)."""
    import ctypes
    from dan_amd import _lib, ops
    g = torch.Generator().manual_seed(3)
    x = torch.randn((N, H, W, Cin), generator=g).to(ops.ACT).to(dev)
    w = (torch.randn((3, 3, Cin, Cout), generator=g) / (9 * Cin) ** 0.5).to(dev)
    b = torch.randn((Cout,), generator=g).to(dev)
    d = ops._desc(N, H, W, Cin, Cout, 3, 3, 1)
    wf, wb = ops.pack_conv_weight(d, w, need_bwd=True)
    co8 = (Cout + 7) // 8 * 8
    dy = torch.zeros((N, H, W, co8), dtype=ops.ACT)
    dy[..., :Cout] = torch.randn((N, H, W, Cout), generator=g).to(ops.ACT)
    dy = dy.to(dev)
    dw = torch.zeros((3, 3, Cin, Cout), dtype=torch.float32, device=dev)
    db = torch.zeros((Cout,), dtype=torch.float32, device=dev)
    side = torch.cuda.Stream()

    def run():
        y = torch.empty((N, H, W, Cout), dtype=ops.ACT, device=dev)
        dx = torch.empty_like(x)
        _lib.call("danhip_conv2d_fwd", ctypes.byref(d), _lib.ptr(x), _lib.ptr(wf), _lib.ptr(b), _lib.ptr(y), _lib.BF16, 1, None, _lib.stream())
        _lib.call("danhip_conv2d_bwd_data", ctypes.byref(d), _lib.ptr(dy), _lib.ptr(wb), _lib.ptr(x), _lib.ptr(dx), 0, _lib.stream())
        torch.cuda.synchronize()
        return y, dx

    try:
        _lib.lib().danhip_set_option(b"halo_b2", 1)
        y2, dx2 = run()
        _lib.lib().danhip_set_option(b"halo_b2", 0)
        for _ in range(12):
            with torch.cuda.stream(side):
                _lib.call("danhip_conv2d_bwd_weight", ctypes.byref(d), _lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), _lib.ptr(db), Cin, _lib.stream())
            y1, dx1 = run()
            assert torch.equal(y1, y2) and torch.equal(dx1, dx2)
    finally:
        _lib.lib().danhip_set_option(b"halo_b2", 0)
        torch.cuda.synchronize()


@pytest.mark.parametrize("N,H,W,Cin,Cout,k", [(16, 40, 40, 512, 512, 3), (3, 37, 40, 256, 128, 3), (2, 48, 48, 256, 256, 1), (4, 20, 40, 1024, 512, 1), (2, 40, 40, 256, 64, (3, 1))])
def test_pointwise_data_gradient_256_wide_tiles_with_epilogue_inputs(N, H, W, Cin, Cout, k, dev):
    """Round 4: conv_pointwise.hip's data gradient WITH a ReLU mask / accumulation can take 256-wide tiles too (option pw_dgrad_ld_bn = 256;
    conv5_x, fc6, the 256-channel pyramid levels: dY is read once instead of Cin / 128 times - measured no faster, so 128 stays the default).
    The K walk of every output is the same in both tilings, so the results must be bit-identical, in all three epilogue modes, and right
    against the fp32 product."""
    import ctypes
    from dan_amd import _lib, ops
    kh, kw = (k, k) if isinstance(k, int) else k
    g = torch.Generator().manual_seed(Cin + Cout + kh)
    x = torch.randn((N, H, W, Cin), generator=g).to(ops.ACT)
    w = (torch.randn((kh, kw, Cin, Cout), generator=g) / (kh * kw * Cin) ** 0.5).to(ops.ACT).float()
    dy = torch.randn((N, H, W, Cout), generator=g).to(ops.ACT)
    old = torch.randn((N, H, W, Cin), generator=g).to(ops.ACT)
    d = ops._desc(N, H, W, Cin, Cout, kh, kw, 1)
    _lib.lib().danhip_set_option(b"pw_dgrad_ld_bn", 256)
    label = _lib.lib().danhip_conv_kernel_label(ctypes.byref(d), 5 | 16).decode()
    _lib.lib().danhip_set_option(b"pw_dgrad_ld_bn", 128)
    assert label.startswith("conv_pointwise_kernel<256, 3, true, true"), label
    _, wb = ops.pack_conv_weight(d, w.to(dev), need_bwd=True)
    xd, dyd, oldd = x.to(dev), dy.to(dev), old.to(dev)
    xr = x.float().requires_grad_(True)
    T.conv2d_same(xr, w, None, stride=1, relu=False).backward(dy.float())
    ref = xr.grad

    def run(masked, acc):
        dx = oldd.clone()
        _lib.call("danhip_conv2d_bwd_data", ctypes.byref(d), _lib.ptr(dyd), _lib.ptr(wb), _lib.ptr(xd) if masked else None, _lib.ptr(dx), acc, _lib.stream())
        torch.cuda.synchronize()
        return dx

    try:
        for masked, acc in ((True, 0), (False, 1), (True, 1)):
            _lib.lib().danhip_set_option(b"pw_dgrad_ld_bn", 128)
            narrow = run(masked, acc)
            _lib.lib().danhip_set_option(b"pw_dgrad_ld_bn", 256)
            wide = run(masked, acc)
            assert torch.equal(narrow, wide), (masked, acc)
            want = torch.where(x.float() > 0, ref, torch.zeros_like(ref)) if masked else ref
            want = want + (old.float() if acc else 0.0)
            err = (wide.float().cpu() - want).abs().max().item()
            assert err <= 2.0 ** -7 * want.abs().max().item() + 2e-2, (masked, acc, err)
    finally:
        _lib.lib().danhip_set_option(b"pw_dgrad_ld_bn", 128)


@pytest.mark.parametrize("N,H,W,Cin,Cout,k", [(4, 160, 160, 128, 256, 3), (2, 96, 128, 256, 256, 3), (4, 160, 160, 256, 256, 1), (2, 80, 80, 2304, 256, 1)])
def test_weight_gradient_one_barrier_form_is_bit_identical_in_the_slab_form(N, H, W, Cin, Cout, k, dev):
    """the default one-barrier K-step of conv_wgrad_rows.hip (3x3) / conv_wgrad_pw.hip (1x1) had no bit-identity check.  With
    `wgrad_slab = 2` the partial tiles leave as plain stores and are combined in a fixed order (no atomics), so dW must be bit-identical (db: fp32 atomics, equal up to order)
    between the two-barrier form (`wgrad_b2 = 1`) and repeated one-barrier runs, also while another stream keeps the chip busy."""
    import ctypes
    from dan_amd import _lib, ops
    g = torch.Generator().manual_seed(4)
    x = torch.randn((N, H, W, Cin), generator=g).to(ops.ACT).to(dev)
    dy = torch.randn((N, H, W, Cout), generator=g).to(ops.ACT).to(dev)
    d = ops._desc(N, H, W, Cin, Cout, k, k, 1)
    L = _lib.lib()
    side = torch.cuda.Stream()
    try:
        L.danhip_set_option(b"wgrad_slab", 2)
        nws = L.danhip_conv2d_bwd_weight_workspace_bytes(ctypes.byref(d))
        assert nws > 0, "shape has no slab form: pick another"
        ws = torch.empty(nws, dtype=torch.uint8, device=dev)
        w0 = (torch.randn((k, k, Cin, Cout), generator=g) / (k * k * Cin) ** 0.5).to(dev)
        wf, _ = ops.pack_conv_weight(d, w0, need_bwd=False)
        b0 = torch.zeros((Cout,), device=dev)

        def run():
            dw = torch.zeros((k, k, Cin, Cout), dtype=torch.float32, device=dev)
            db = torch.zeros((Cout,), dtype=torch.float32, device=dev)
            _lib.call("danhip_conv2d_bwd_weight_ws", ctypes.byref(d), _lib.ptr(x), _lib.ptr(dy), _lib.ptr(dw), _lib.ptr(db), Cin, _lib.ptr(ws), nws, _lib.stream())
            torch.cuda.synchronize()
            return dw, db

        L.danhip_set_option(b"wgrad_b2", 1)
        dw2, db2 = run()
        L.danhip_set_option(b"wgrad_b2", 0)
        assert dw2.abs().max().item() > 0
        for _ in range(10):
            with torch.cuda.stream(side):
                y = torch.empty((N, H, W, Cout), dtype=ops.ACT, device=dev)
                _lib.call("danhip_conv2d_fwd", ctypes.byref(d), _lib.ptr(x), _lib.ptr(wf), _lib.ptr(b0), _lib.ptr(y), _lib.BF16, 1, None, _lib.stream())
            dw1, db1 = run()
            assert torch.equal(dw1, dw2)
            # (the bias gradient stays a handful of fp32 atomics per channel in both forms: equal up to summation order)
            assert (db1 - db2).abs().max().item() <= 1e-5 * db2.abs().max().item()
    finally:
        L.danhip_set_option(b"wgrad_slab", 1)
        L.danhip_set_option(b"wgrad_b2", 0)
        torch.cuda.synchronize()


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 64, 96, 64, 64), (4, 160, 160, 128, 128), (2, 64, 64, 128, 128), (1, 45, 67, 128, 256), (2, 40, 40, 512, 512)])
def test_pool_only_inference_conv_equals_conv_then_pool(N, H, W, Cin, Cout, dev):
    """Inference (round 4): conv1_2 / conv2_2 feed nothing but their 2x2 pool, so danhip_conv2d_fwd_pool(y = NULL) never writes the
    full-resolution map where the kernel pools in its epilogue (the 64 -> 64 kernel; the halo kernel's lean epilogue: a zero-length store
    descriptor).  The pooled map must be bit-identical to conv + pool; shapes whose pool is a separate kernel take the ordinary route."""
    from dan_amd import ops
    g = torch.Generator().manual_seed(N * 7 + Cin)
    x = torch.randn((N, H, W, Cin), generator=g).to(ops.ACT).to(dev)
    w = (torch.randn((3, 3, Cin, Cout), generator=g) / (9 * Cin) ** 0.5).to(dev)
    b = torch.randn((Cout,), generator=g).to(dev)
    with torch.no_grad():
        want = ops.max_pool_2x2(ops.conv2d(x, w, b, relu=True, pool=True))
        got = ops.max_pool_2x2(ops.conv2d(x, w, b, relu=True, pool=True, pool_only=True))
    torch.cuda.synchronize()
    assert got.shape == (N, (H + 1) // 2, (W + 1) // 2, Cout)
    if N * H * W >= 65536 or Cin == 64:          # both routes on the same kernel: bit-identical
        assert torch.equal(got, want)
    else:                                        # (the small maps' ordinary route may split K over workgroups: another fp32 summation order)
        assert (got.float() - want.float()).abs().max().item() <= 2.0 ** -7 * want.float().abs().max().item()
    # with a gradient tracked the tensor keeps its shape (its DATA may be absent: next test)
    xg = x.clone().requires_grad_(True)
    y = ops.conv2d(xg, w, b, relu=True, pool=True, pool_only=True)
    assert y.shape == (N, H, W, Cout)


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(2, 64, 96, 64, 64), (4, 160, 160, 128, 128), (1, 45, 67, 128, 256), (2, 40, 40, 512, 512)])
def test_pool_only_training_skips_the_full_resolution_stores_and_changes_no_gradient(N, H, W, Cin, Cout, dev, monkeypatch):
    """Training (round 5): conv1_2 / conv2_2 (net/sfd_net.py:128-135) feed nothing but their 2x2 pool, and backward reaches them only through
    the pool's arg-max codes with the gradient already masked at the pooled level - so with the direct gradient hand-off the forward call
    gets y = NULL (ops.POOL_ONLY_TRAIN): same pooled map bit for bit, same dX / dW / db as the form that writes y (weight gradients up
    to the order of their fp32 partial sums); the declared-but-unwritten tensor cannot be read by any op of this package."""
    from dan_amd import ops
    g = torch.Generator().manual_seed(N * 11 + Cin)
    x = torch.randn((N, H, W, Cin), generator=g).to(ops.ACT).to(dev)
    w = (torch.randn((3, 3, Cin, Cout), generator=g) / (9 * Cin) ** 0.5).to(dev)
    b = torch.randn((Cout,), generator=g).to(dev)
    w2 = (torch.randn((3, 3, Cout, 64), generator=g) / (9 * Cout) ** 0.5).to(dev)
    b2 = torch.randn((64,), generator=g).to(dev)
    gz = torch.randn((N, (H + 1) // 2, (W + 1) // 2, 64), generator=g).to(ops.ACT).to(dev)

    def run(flag):
        monkeypatch.setattr(ops, "POOL_ONLY_TRAIN", flag)
        xg, wg, bg = x.clone().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
        y = ops.conv2d(xg, wg, bg, relu=True, pool=True, pool_only=True)
        p = ops.max_pool_2x2(y)
        z = ops.conv2d(p, w2, b2, relu=True)
        z.backward(gz)
        torch.cuda.synchronize()
        return y, p.detach().clone(), xg.grad.clone(), wg.grad.clone(), bg.grad.clone()

    y0, p0, dx0, dw0, db0 = run(False)
    y1, p1, dx1, dw1, db1 = run(True)
    assert y0.is_contiguous() and tuple(y1.shape) == tuple(y0.shape)
    skipped = not y1.is_contiguous()
    d = ops._desc(N, H, W, Cin, Cout, 3, 3, 1)
    fused_in_epilogue = bool(ops._lib.lib().danhip_conv2d_fwd_pool_only(ctypes.byref(d))) and ops._conv_scratch(d, 0, dev)[1] == 0
    assert skipped == fused_in_epilogue, (skipped, fused_in_epilogue)
    assert torch.equal(p0, p1) and torch.equal(dx0, dx1)
    assert (dw0 - dw1).abs().max().item() <= 1e-4 * dw0.abs().max().item() and (db0 - db1).abs().max().item() <= 1e-4 * db0.abs().max().item()
    if skipped:
        with pytest.raises(AssertionError):              # nothing can read the unwritten map: ptr() refuses the zero-stride view
            ops.conv2d(y1.detach(), w2[:, :, :Cout], b2, relu=True)


@pytest.mark.parametrize("N,H,W,cin_real", [(2, 37, 131, 3), (1, 8, 64, 3), (3, 64, 100, 4), (1, 5, 1, 1), (2, 96, 96, 3), (1, 203, 331, 3)])
def test_first_layer_weight_gradient_kernel_vs_the_oracle_and_the_general_kernel(N, H, W, cin_real, dev):
    """conv_wgrad_c8.hip (round 5): conv1_1's weight and bias gradient (net/sfd_net.py:128; 3x3 / 'same', the image padded to 8 channels) with
    both operands staged once and the nine taps as address offsets of one halo patch, against the fp32 oracle gradient (autograd through
    oracle.tf_ops.conv2d_same on the 16-bit-rounded operands: what remains is the fp32 summation order, 1e-3 of the gradient's scale) and
    against the general kernel it replaces (option "wgrad_c8" = 0).  Ragged sizes: tiles of 8 x 64 pixels with zero-filled remainders;
    accumulation into a non-zero dW / db (the flat gradient buffer's += semantics)."""
    from dan_amd import _lib, ops
    g = torch.Generator().manual_seed(N * 13 + H)
    x = torch.zeros((N, H, W, 8))
    x[..., :cin_real] = torch.randn((N, H, W, cin_real), generator=g)
    x = x.to(ops.ACT)
    dy = torch.randn((N, H, W, 64), generator=g).to(ops.ACT)
    w = torch.zeros((3, 3, cin_real, 64), requires_grad=True)
    b = torch.zeros((64,), requires_grad=True)
    y = T.conv2d_same(x[..., :cin_real].float(), w, b)
    (y * dy.float()).sum().backward()
    d = ops._desc(N, H, W, 8, 64, 3, 3, 1)
    xd, dyd = x.to(dev), dy.to(dev)
    L = _lib.lib()
    outs = []
    for new in (1, 0):
        assert L.danhip_set_option(b"wgrad_c8", new) == 0
        dw = torch.full((3, 3, cin_real, 64), 0.5, dtype=torch.float32, device=dev)
        db = torch.full((64,), -0.25, dtype=torch.float32, device=dev)
        try:
            assert (L.danhip_conv_wgrad_kernel_label(ctypes.byref(d)) == b"conv_wgrad_c8_kernel") == bool(new)
            _lib.call("danhip_conv2d_bwd_weight", ctypes.byref(d), _lib.ptr(xd), _lib.ptr(dyd), _lib.ptr(dw), _lib.ptr(db), cin_real, _lib.stream())
            torch.cuda.synchronize()
        finally:
            L.danhip_set_option(b"wgrad_c8", 1)
        outs.append((dw.cpu() - 0.5, db.cpu() + 0.25))
    sw, sb = w.grad.abs().max().item() + 1e-6, b.grad.abs().max().item() + 1e-6
    for dw, db in outs:
        assert (dw - w.grad).abs().max().item() <= 1e-3 * sw + 2e-6 * N * H * W ** 0.5      # (+ the 0.5 start value's fp32 rounding of large sums)
        assert (db - b.grad).abs().max().item() <= 1e-3 * sb + 2e-6 * N * H * W ** 0.5
    assert (outs[0][0] - outs[1][0]).abs().max().item() <= 1e-3 * sw + 2e-6 * N * H * W ** 0.5


@pytest.mark.parametrize("shape", [(2, 24, 64), (1, 30, 62), (9, 64, 128), (3, 40, 96), (2, 128, 128)])
def test_second_layer_data_gradient_with_the_first_layers_weight_gradient_folded_in(shape, dev):
    """danhip_conv2d_bwd_data_bits_first (conv_halo_c64.hip FUSE8, round 6): conv1_2's data gradient that keeps its output tiles in LDS and folds
    conv1_1's weight / bias gradient in, against the two calls it replaces on the SAME inputs — danhip_conv2d_bwd_data_bits (dX stored in 16
    bits) followed by danhip_conv2d_bwd_weight of the first layer on that dX.  Both consume the identical 16-bit dX values; what differs is the
    fp32 summation order: 1e-4 of the gradient's max-norm.  Ragged tile edges, several tiles per workgroup, more tiles than workgroups; the
    gradients are ADDED to what the sinks hold."""
    import ctypes
    from dan_amd import ops
    from dan_amd._lib import call, lib, ptr, stream
    N, H, W = shape
    g = torch.Generator().manual_seed(sum(shape) + 1)
    d1 = ops._desc(N, H, W, 8, 64, 3, 3, 1)
    d2 = ops._desc(N, H, W, 64, 64, 3, 3, 1)
    assert lib().danhip_conv2d_bwd_data_first_supported(ctypes.byref(d2)) == 1
    x0 = torch.zeros((N, H, W, 8), dtype=torch.bfloat16)
    x0[..., :3] = torch.randn((N, H, W, 3), generator=g).to(torch.bfloat16)
    x0d = x0.to(dev)
    y1 = torch.randn((N, H, W, 64), generator=g).to(torch.bfloat16).to(dev)          # conv1_1's output: only its sign pattern matters here
    bits = torch.empty((N * H * W, 8), dtype=torch.uint8, device=dev)
    call("danhip_relu_bits", ptr(y1), ptr(bits), N * H * W, 64, stream())
    w2 = (torch.randn((3, 3, 64, 64), generator=g) / 576 ** 0.5).to(dev)
    _, wb = ops.pack_conv_weight(d2, w2, need_bwd=True)
    dy = torch.randn((N, H, W, 64), generator=g).to(torch.bfloat16).to(dev)
    # the two calls it replaces
    dx = torch.empty((N, H, W, 64), dtype=torch.bfloat16, device=dev)
    call("danhip_conv2d_bwd_data_bits", ctypes.byref(d2), ptr(dy), ptr(wb), ptr(bits), ptr(dx), 0, stream())
    seed_w = torch.randn((3, 3, 3, 64), generator=g).to(dev)
    seed_b = torch.randn((64,), generator=g).to(dev)
    dw_ref, db_ref = seed_w.clone(), seed_b.clone()
    call("danhip_conv2d_bwd_weight", ctypes.byref(d1), ptr(x0d), ptr(dx), ptr(dw_ref), ptr(db_ref), 3, stream())
    # the folded call
    dw, db = seed_w.clone(), seed_b.clone()
    call("danhip_conv2d_bwd_data_bits_first", ctypes.byref(d2), ptr(dy), ptr(wb), ptr(bits), ptr(x0d), 3, ptr(dw), ptr(db), stream())
    torch.cuda.synchronize()
    assert (dw_ref - seed_w).abs().max().item() > 0
    for got, want, what in ((dw, dw_ref, "dW"), (db, db_ref, "db")):
        err = (got - want).abs().max().item()
        assert torch.isfinite(got).all() and err <= 1e-4 * want.abs().max().item(), (what, err, want.abs().max().item())
    # no bias sink: allowed
    dw2 = seed_w.clone()
    call("danhip_conv2d_bwd_data_bits_first", ctypes.byref(d2), ptr(dy), ptr(wb), ptr(bits), ptr(x0d), 3, ptr(dw2), None, stream())
    torch.cuda.synchronize()
    assert (dw2 - dw_ref).abs().max().item() <= 1e-4 * dw_ref.abs().max().item()
    d3 = ops._desc(1, 30, 47, 64, 64, 3, 3, 1)      # odd width: no bit-mask form, no folded form
    assert lib().danhip_conv2d_bwd_data_first_supported(ctypes.byref(d3)) == 0


def test_training_step_with_and_without_the_folded_first_layer_gradient(dev):
    """The S3FD trainer's step with ops.FUSE_FIRST_WGRAD on (default) and off: every variable's gradient agrees (fp32 summation order on
    conv1_1's kernel and bias, bit-identical elsewhere), the first layer's gradient is non-zero, and with the fold the first layer's own
    weight-gradient kernel is not launched (ops.PROFILE sees no first-layer weight gradient)."""
    from dan_amd import ops, synthetic
    from dan_amd.train_sfd import AnchorConfig, SFDModel, SFDTrainer
    B, S = 2, 128
    imgs = synthetic.make_images(B, S, S, dev, seed=3)
    model = SFDModel(device=dev, seed=5)
    anchors = AnchorConfig(S, S, dev)
    loc_t, cls_t, _ = anchors.encode_batch(synthetic.make_gt_boxes(B, S, S, seed=4, max_faces=5))
    tr = SFDTrainer(model)
    w0 = tr.flat.w.clone()
    grads = {}
    for fuse in (True, False):
        tr.flat.w.copy_(w0); tr.flat.v.zero_(); tr.step_no = 0
        ops.WEIGHT_EPOCH += 1
        ops.repack_all()
        with ops.use_context(ops.OpsContext(FUSE_FIRST_WGRAD=fuse)):
            ops.PROFILE = {}
            tr.train_step(imgs, loc_t, cls_t)
            prof, ops.PROFILE = ops.PROFILE, None
        torch.cuda.synchronize()
        grads[fuse] = tr.flat.g.clone()
        first_wgrad = [k for k in prof if "wgrad_c8" in k]
        assert (len(first_wgrad) == 0) == fuse, (fuse, list(prof))
    names = dict(zip(tr.flat.names, zip(tr.flat.starts, tr.flat.sizes)))
    first = [n for n in tr.flat.names if "conv1_1" in n]
    assert len(first) == 2
    for n, (s0, sz) in names.items():
        a, b = grads[True][s0:s0 + sz], grads[False][s0:s0 + sz]
        if n in first:
            assert b.abs().max().item() > 0
            assert (a - b).abs().max().item() <= 2e-4 * b.abs().max().item(), n
        else:
            assert (a - b).abs().max().item() <= 2e-5 * max(b.abs().max().item(), 1e-20), n      # (fp32 atomics order of the weight gradients)
