"""The north-star bound "eval_dan.py box outputs within 1e-4 of the reference on identical weights/inputs" at 16-bit MFMA rates (VERDICT r5
item 5): model.precision = "split" runs the fp32 evaluation graphs with every convolution as a SPLIT-OPERAND product on the fp16 MFMA
(csrc/split_infer.hip: x = hi + lo in two IEEE-half limbs, hi.hi + lo.hi + hi.lo with exact products and fp32 accumulation, laid out as an
ordinary convolution over 3C input channels).  Same cases, same oracle, SAME tolerances as the fp32 path's tests (tests/test_eval_f32_gpu.py):
logits 1e-4 of their scale, every decoded box coordinate within 1e-4 * max(1, |ref|) px, DAN's routed boxes likewise where the discrete
routing decision agrees (<= 0.1 % may flip)."""
import pytest
import torch

from test_eval_f32_gpu import SIZES, dan_eval_case, single_stage_case

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("h,w", SIZES)
@pytest.mark.parametrize("which", ["sfd", "pb"])
def test_single_stage_eval_boxes_split(which, h, w, dev):
    single_stage_case(which, h, w, dev, "split")


@pytest.mark.parametrize("h,w", SIZES)
@pytest.mark.parametrize("deform", [False, True])
def test_dan_eval_boxes_split(deform, h, w, dev):
    dan_eval_case(deform, h, w, dev, precision="split")


@pytest.mark.parametrize("M,C", [(1000, 64), (777, 3), (513, 72), (64, 1024)])
def test_split3_round_trip_and_limb_layout(M, C, dev):
    """danhip_split3_f32 / danhip_unsplit3_f32: X3 = [hi | lo | hi | 0], hi = half(x), lo = half(x - hi); hi + lo recovers x to 2^-22
    relative (or half's subnormal step 2^-24 absolute), the padding columns are zero."""
    from dan_amd import _lib, ops
    g = torch.Generator().manual_seed(M + C)
    x = (torch.randn((1, 1, M, C), generator=g) * torch.logspace(-3, 3, C).view(1, 1, 1, C)).to(dev)
    x3 = ops.split3(x)
    C3 = (3 * C + 7) // 8 * 8
    assert x3.shape == (1, 1, M, C3) and x3.dtype == torch.float16
    hi = x.half()
    lo = (x - hi.float()).half()
    assert torch.equal(x3[..., :C], hi) and torch.equal(x3[..., C:2 * C], lo) and torch.equal(x3[..., 2 * C:3 * C], hi)
    assert (x3[..., 3 * C:] == 0).all()
    y = torch.empty_like(x)
    _lib.call("danhip_unsplit3_f32", _lib.ptr(x3), _lib.ptr(y), M, C, C3, _lib.stream())
    err = (y - x).abs()
    assert (err <= x.abs() * 2.0 ** -21 + 2.0 ** -24).all(), err.max().item()


@pytest.mark.parametrize("N,H,W,C", [(2, 64, 96, 64), (1, 33, 47, 128)])
def test_maxpool_on_the_split_layout_equals_the_fp32_pool(N, H, W, C, dev):
    from dan_amd import _lib, ops
    g = torch.Generator().manual_seed(H * W)
    x = torch.randn((N, H, W, C), generator=g).to(dev)
    x3 = ops.split3(x)
    xq = x3[..., :C].float() + x3[..., C:2 * C].float()            # what the limbs carry
    want = ops.max_pool_2x2(xq.contiguous())
    y3 = torch.empty((N, (H + 1) // 2, (W + 1) // 2, 3 * C), dtype=torch.float16, device=dev)
    _lib.call("danhip_maxpool2x2_split3", _lib.ptr(x3), _lib.ptr(y3), N, H, W, C, _lib.stream())
    got = y3[..., :C].float() + y3[..., C:2 * C].float()
    assert torch.equal(y3[..., :C], y3[..., 2 * C:])
    assert (got - want).abs().max().item() <= 2.0 ** -21 * want.abs().max().item()


@pytest.mark.parametrize("N,H,W,Cin,Cout,k,stride,relu", [
    (2, 96, 96, 64, 128, 3, 1, True),           # halo kernel, 192 input channels
    (1, 125, 167, 3, 64, 3, 1, True),           # the image: 9 limb channels padded to 16
    (2, 150, 150, 256, 256, 1, 1, True),        # pointwise
    (4, 65, 63, 128, 256, 3, 2, False),         # stride 2
    (1, 40, 40, 512, 6, 3, 1, False),           # a detection head (ragged Cout, fp32 output as always)
])
def test_split_conv_vs_the_fp32_oracle(N, H, W, Cin, Cout, k, stride, relu, dev):
    """ops.conv2d with SPLIT_EVAL against oracle.tf_ops.conv2d_same in fp32: 4e-6 of the output scale (the fp32 kernel's own bound is 1e-5;
    a plain half convolution sits at ~1e-3)."""
    from dan_amd import ops
    from oracle import tf_ops as T
    g = torch.Generator().manual_seed(N * 31 + Cin)
    x = torch.randn((N, H, W, Cin), generator=g)
    w = torch.randn((k, k, Cin, Cout), generator=g) / (k * k * Cin) ** 0.5
    b = torch.randn((Cout,), generator=g)
    want = T.conv2d_same(x, w, b, stride=stride, relu=relu)
    with ops.use_context(ops.OpsContext(SPLIT_EVAL=True)), torch.no_grad():
        y = ops.conv2d(x.to(dev), w.to(dev), b.to(dev), stride=stride, relu=relu)
        if Cout % 8 == 0:                      # the kernel's epilogue wrote the NEXT convolution's limb layout: a half view of the hi limbs + the map
            assert y.dtype == torch.float16 and y.shape[-1] == Cout and y._dh_split3.shape[-1] == 3 * Cout
            assert torch.equal(y._dh_split3[..., :Cout], y._dh_split3[..., 2 * Cout:])
        got = ops._f32_in(y).cpu()
    assert got.dtype == torch.float32 and torch.isfinite(got).all()
    assert (got - want).abs().max().item() <= 4e-6 * want.abs().max().item(), (got - want).abs().max().item() / want.abs().max().item()


def test_two_split_convs_chained_through_the_limb_layout_and_a_pool(dev):
    """conv -> (epilogue limb layout) -> conv -> pool on the limb layout -> conv with fp32 output, against the fp32 oracle chain: the
    limbs handed from epilogue to operand lose nothing beyond 2^-22 per value."""
    from dan_amd import ops
    from oracle import tf_ops as T
    g = torch.Generator().manual_seed(5)
    x = torch.randn((2, 64, 96, 64), generator=g)
    ws = [torch.randn((3, 3, ci, co), generator=g) / (9 * ci) ** 0.5 for ci, co in ((64, 128), (128, 128), (128, 8))]
    bs = [0.1 * torch.randn((w.shape[-1],), generator=g) for w in ws]
    want = T.conv2d_same(x, ws[0], bs[0], stride=1, relu=True)
    want = T.conv2d_same(want, ws[1], bs[1], stride=1, relu=True)
    want = T.max_pool_2x2_same(want)
    want = T.conv2d_same(want, ws[2], bs[2], stride=1, relu=False)
    with ops.use_context(ops.OpsContext(SPLIT_EVAL=True)), torch.no_grad():
        y = ops.conv2d(x.to(dev), ws[0].to(dev), bs[0].to(dev), relu=True)
        y = ops.conv2d(y, ws[1].to(dev), bs[1].to(dev), relu=True)
        assert ops._is_limbs(y)
        y = ops.max_pool_2x2(y)
        assert ops._is_limbs(y) and y.shape == (2, 32, 48, 128)
        y = ops.conv2d(y, ws[2].to(dev), bs[2].to(dev), relu=False, out_f32=True)
    assert y.dtype == torch.float32
    assert (y.cpu() - want).abs().max().item() <= 1e-5 * want.abs().max().item()


def test_l2norm_on_the_limb_layout_equals_the_fp32_chain(dev):
    """danhip_l2norm_split3 against unsplit -> danhip_l2norm_fwd_f32 -> split (same lane-strided sum, shuffle reduction and multiply order; the
    two compilations may contract differently): equal to the limbs' own precision, 2^-20 of the map's maximum."""
    from dan_amd import ops
    g = torch.Generator().manual_seed(3)
    x = (torch.randn((2, 40, 56, 256), generator=g) * 3).to(dev)
    gamma = (10 + torch.randn((256,), generator=g)).to(dev)
    xv = ops._limb_view(ops.split3(x), 256)
    got = ops.l2_normalize(xv, gamma)
    assert ops._is_limbs(got)
    want = ops.l2_normalize(ops.unsplit3(xv), gamma)
    assert torch.equal(got._dh_split3[..., :256], got._dh_split3[..., 512:])
    assert (ops.unsplit3(got) - want).abs().max().item() <= 2.0 ** -20 * want.abs().max().item()


def test_a_split_conv_larger_than_2_pow_31_elements_runs_in_batch_slices(dev):
    """The deformable GEMM's operand at batch 16 (27 x 256 limb channels at 160 x 160) exceeds the 16-bit kernels' 32-bit element offsets:
    _conv2d_split slices the batch.  Here: 1 x 1 over 6144 channels (18432 limb channels), 6 images of 160 x 160 = 2.8e9 input elements."""
    from dan_amd import ops
    from oracle import tf_ops as T
    g = torch.Generator().manual_seed(4)
    x = torch.randn((6, 160, 160, 6144), generator=g, dtype=torch.float32)
    w = torch.randn((1, 1, 6144, 8), generator=g) / 6144 ** 0.5
    want = torch.einsum("nhwc,co->nhwo", x.double(), w[0, 0].double()).float()
    with ops.use_context(ops.OpsContext(SPLIT_EVAL=True)), torch.no_grad():
        got = ops._f32_in(ops.conv2d(x.to(dev), w.to(dev), None, relu=False)).cpu()
    # (K = 6144 products accumulated in fp32: sqrt(K) * 2^-24 = 4.7e-6 of the terms' scale on top of the limbs' 2^-22)
    assert got.shape == want.shape and (got - want).abs().max().item() <= 2e-5 * want.abs().max().item()
