"""DeformConvOp attributes no graph of the reference uses (cpp/Deform/deform_conv.cc:51-167: padding 'VALID', num_groups > 1,
data_format 'NCHW'), served by exact compositions of the SAME / one-group / NHWC kernels (utility/custom_op.py:deform_conv_op) and checked
against the oracle restatement with the same attributes (oracle/deform.py)."""
import pytest
import torch

from oracle import deform as OD

pytestmark = pytest.mark.gpu
TOL_F, TOL_G = 2.0 ** -6, 2.0 ** -5          # as tests/test_deform_gpu.py: 2 and 3 roundings to 16 bits on the way


def _rand(shape, g, scale=1.0):
    return (torch.randn(shape, generator=g) * scale).to(torch.bfloat16)


@pytest.mark.parametrize("dil", [1, 2])
def test_valid_padding_matches_the_oracle_forward_and_backward(dil, dev):
    from dan_amd.utility import custom_op
    g = torch.Generator().manual_seed(11 + dil)
    N, H, W, C, Cout, dg = 2, 13, 10, 64, 32, 2
    x = _rand((N, H, W, C), g)
    w = (torch.randn((Cout, C, 3, 3), generator=g) / (9 * C) ** 0.5).to(torch.bfloat16).float()
    off = _rand((N, H - 2, W - 2, dg * 18), g, 1.2)
    xr, offr = x.float().permute(0, 3, 1, 2), off.float().permute(0, 3, 1, 2)
    ref = OD.deform_conv_forward(xr, w, offr, 1, dil, dg, padding="VALID")
    dy = _rand(tuple(ref.permute(0, 2, 3, 1).shape), g)
    dxr, dwr, dor = OD.deform_conv_backward(xr, w, offr, dy.float().permute(0, 3, 1, 2), 1, dil, dg, padding="VALID")
    xd, wd, od = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True), off.to(dev).requires_grad_(True)
    y = custom_op.deform_conv_op(xd, wd, od, [1, 1, dil, dil], "VALID", [1, 1, 1, 1], 1, dg)
    assert tuple(y.shape) == (N, H - 2, W - 2, Cout)
    y.backward(dy.to(dev))
    want = ref.permute(0, 2, 3, 1)
    assert (y.float().cpu() - want).abs().max().item() <= TOL_F * want.abs().max().item() + 1e-3
    for name, got, ref_g in (("dx", xd.grad.float().cpu(), dxr.permute(0, 2, 3, 1)), ("dw", wd.grad.cpu(), dwr),
                             ("doffset", od.grad.float().cpu(), dor.permute(0, 2, 3, 1))):
        scale = ref_g.abs().max().item() + 1e-6
        assert (got - ref_g).abs().max().item() <= TOL_G * scale + 2e-3, name
    with pytest.raises(ValueError):
        custom_op.deform_conv_op(xd, wd, od, [1, 1, 1, 1], "VALID", [1, 1, 2, 2], 1, dg)


@pytest.mark.parametrize("G,dg", [(2, 4), (4, 2), (2, 2)])
def test_channel_groups_match_the_oracle_forward_and_autograd_backward(G, dg, dev):
    from dan_amd.utility import custom_op
    g = torch.Generator().manual_seed(G * 10 + dg)
    N, H, W, C, Cout = 1, 9, 11, 256, 64
    x = _rand((N, H, W, C), g)
    w = (torch.randn((Cout, C // G, 3, 3), generator=g) / (9 * C / G) ** 0.5).to(torch.bfloat16).float()
    off = _rand((N, H, W, dg * 18), g, 0.8)
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    ref = OD.deform_conv_forward(xr, wr, off.float().permute(0, 3, 1, 2), 1, 1, dg, num_groups=G)
    dy = _rand((N, H, W, Cout), g)
    ref.backward(dy.float().permute(0, 3, 1, 2))                     # (the oracle's im2col is differentiable in x and w; offsets: see the one-group tests)
    xd, wd, od = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True), off.to(dev).requires_grad_(True)
    y = custom_op.deform_conv_op(xd, wd, od, [1, 1, 1, 1], "SAME", [1, 1, 1, 1], G, dg)
    y.backward(dy.to(dev))
    want = ref.detach().permute(0, 2, 3, 1)
    assert (y.float().cpu() - want).abs().max().item() <= TOL_F * want.abs().max().item() + 1e-3
    for name, got, ref_g in (("dx", xd.grad.float().cpu(), xr.grad.permute(0, 2, 3, 1)), ("dw", wd.grad.cpu(), wr.grad)):
        scale = ref_g.abs().max().item() + 1e-6
        assert (got - ref_g).abs().max().item() <= TOL_G * scale + 2e-3, name
    assert od.grad is not None and od.grad.abs().max().item() > 0
    with pytest.raises(ValueError):
        custom_op.deform_conv_op(xd, wd[:, :C // G - 8], od, [1, 1, 1, 1], "SAME", [1, 1, 1, 1], G, dg)


def test_nchw_is_the_nhwc_op_transposed(dev):
    from dan_amd.utility import custom_op
    g = torch.Generator().manual_seed(5)
    N, H, W, C, Cout, dg = 1, 8, 12, 64, 64, 1
    x, off = _rand((N, H, W, C), g).to(dev), _rand((N, H, W, dg * 18), g, 0.6).to(dev)
    w = (torch.randn((Cout, C, 3, 3), generator=g) / (9 * C) ** 0.5).to(dev)
    a = custom_op.deform_conv_op(x, w, off, [1, 1, 1, 1], "SAME", [1, 1, 1, 1], 1, dg)
    b = custom_op.deform_conv_op(x.permute(0, 3, 1, 2).contiguous(), w, off.permute(0, 3, 1, 2).contiguous(), [1, 1, 1, 1], "SAME", [1, 1, 1, 1], 1, dg,
                                 data_format="NCHW")
    assert tuple(b.shape) == (N, Cout, H, W) and torch.equal(b.permute(0, 2, 3, 1), a)
