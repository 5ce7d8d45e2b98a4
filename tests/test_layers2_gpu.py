"""Parity of the LFPN / context-module layer kernels (through the C ABI) against the CPU oracle's TF-semantics ops:
legacy resize_bilinear + lateral add, padding-excluding 2x2/1 average pool, batch normalisation.
bf16 storage: inputs are rounded to bf16 first, the oracle computes in fp32 on those values; tolerance = one bf16
rounding of the output (2^-8 relative) + accumulation slack, stated per test."""
import pytest
import torch

from oracle import tf_ops as T

pytestmark = pytest.mark.gpu


def _bf(t):
    return t.to(torch.bfloat16)


@pytest.mark.parametrize("shape,out", [((2, 20, 20, 64), (40, 40)), ((1, 5, 7, 16), (10, 14)), ((1, 6, 5, 8), (11, 9)), ((1, 8, 8, 8), (8, 8))])
def test_resize_bilinear_add(shape, out, dev):
    from dan_amd import ops
    g = torch.Generator().manual_seed(0)
    up = _bf(torch.randn(shape, generator=g))
    lat = _bf(torch.randn((shape[0], out[0], out[1], shape[3]), generator=g))
    upr = up.float().requires_grad_(True)
    latr = lat.float().requires_grad_(True)
    ref = latr + T.resize_bilinear_legacy(upr, out[0], out[1])
    dy = _bf(torch.randn(ref.shape, generator=g))
    ref.backward(dy.float())
    upd = up.to(dev).requires_grad_(True)
    latd = lat.to(dev).requires_grad_(True)
    y = ops.resize_bilinear_add(upd, latd)
    y.backward(dy.to(dev))
    torch.cuda.synchronize()
    tol = 2.0 ** -7 * ref.abs().max().item()
    assert (y.float().cpu() - ref.detach()).abs().max().item() <= tol
    assert torch.equal(latd.grad.cpu(), dy)                                    # lateral gradient is the upstream gradient itself
    gtol = 2.0 ** -7 * upr.grad.abs().max().item() + 1e-3
    assert (upd.grad.float().cpu() - upr.grad).abs().max().item() <= gtol
    # plain resize (no lateral)
    y2 = ops.resize_bilinear_add(up.to(dev), None, out)
    ref2 = T.resize_bilinear_legacy(up.float(), out[0], out[1])
    assert (y2.float().cpu() - ref2).abs().max().item() <= 2.0 ** -7 * ref2.abs().max().item()


@pytest.mark.parametrize("shape", [(2, 16, 16, 64), (1, 5, 7, 8), (1, 1, 1, 8), (1, 2, 9, 16)])
def test_avgpool_2x2_s1_same(shape, dev):
    from dan_amd import ops
    g = torch.Generator().manual_seed(1)
    x = _bf(torch.randn(shape, generator=g))
    xr = x.float().requires_grad_(True)
    ref = T.avg_pool_2x2_s1_same(xr)
    dy = _bf(torch.randn(ref.shape, generator=g))
    ref.backward(dy.float())
    xd = x.to(dev).requires_grad_(True)
    y = ops.avg_pool_2x2_s1(xd)
    y.backward(dy.to(dev))
    torch.cuda.synchronize()
    assert (y.float().cpu() - ref.detach()).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item() + 1e-6
    assert (xd.grad.float().cpu() - xr.grad).abs().max().item() <= 2.0 ** -7 * xr.grad.abs().max().item() + 1e-6


@pytest.mark.parametrize("shape,relu", [((2, 12, 12, 64), False), ((3, 7, 5, 16), True), ((1, 4, 4, 256), False)])
def test_batch_norm(shape, relu, dev):
    from dan_amd import ops
    g = torch.Generator().manual_seed(2)
    C = shape[-1]
    x = _bf(torch.randn(shape, generator=g) * 2.0 + 0.5)
    gamma = torch.rand(C, generator=g) + 0.5
    beta = torch.randn(C, generator=g) * 0.1
    xr = x.float().requires_grad_(True)
    gr = gamma.clone().requires_grad_(True)
    br = beta.clone().requires_grad_(True)
    y_ref, mean_ref, var_ref = T.batch_norm_train(xr, gr, br, 1e-5)
    if relu:
        y_ref = torch.relu(y_ref)
    dy = _bf(torch.randn(shape, generator=g))
    y_ref.backward(dy.float())
    xd = x.to(dev).requires_grad_(True)
    gd = gamma.to(dev).requires_grad_(True)
    bd = beta.to(dev).requires_grad_(True)
    mm = torch.zeros(C, device=dev)
    mv = torch.ones(C, device=dev)
    y = ops.batch_norm_train(xd, gd, bd, mm, mv, eps=1e-5, momentum=0.997, relu=relu)
    y.backward(dy.to(dev))
    torch.cuda.synchronize()
    assert (y.float().cpu() - y_ref.detach()).abs().max().item() <= 2.0 ** -7 * y_ref.abs().max().item() + 1e-3
    # moving averages: moving*m + batch*(1-m); the fused TF1 path feeds the Bessel-corrected variance to moving_variance
    assert torch.allclose(mm.cpu(), mean_ref.detach() * (1 - 0.997), atol=1e-5)
    count = float(x.numel() // C)
    assert torch.allclose(mv.cpu(), T.batch_norm_moving_variance(torch.ones(C), var_ref.detach(), count, 0.997), atol=1e-6)
    assert not torch.allclose(mv.cpu(), 0.997 + var_ref.detach() * (1 - 0.997), atol=1e-7) or count > 1e5
    if not relu:                                        # (with ReLU the mask is taken on bf16-rounded outputs: covered by the forward check)
        for got, want, name in ((xd.grad.float().cpu(), xr.grad, "dx"), (gd.grad.cpu(), gr.grad, "dgamma"), (bd.grad.cpu(), br.grad, "dbeta")):
            assert (got - want).abs().max().item() <= 2.0 ** -6 * want.abs().max().item() + 2e-3, name
    yi = ops.batch_norm_infer(x.to(dev), gamma.to(dev), beta.to(dev), mean_ref.detach().to(dev), var_ref.detach().to(dev), eps=1e-5, relu=relu)
    assert (yi.float().cpu() - y_ref.detach()).abs().max().item() <= 2.0 ** -7 * y_ref.abs().max().item() + 1e-3


def test_backbone_bn_surface(dev):
    """conv_bn_relu / bn_relu / conv_bn of VGG16Backbone (net/sfd_net.py:91-119) against the oracle's conv + batch norm."""
    from dan_amd.net import sfd_net
    g = torch.Generator().manual_seed(4)
    x = _bf(torch.randn((2, 10, 12, 64), generator=g))
    bb = sfd_net.VGG16Backbone("channels_last", variables=sfd_net.VariableStore(device=dev, seed=1))
    y = bb.conv_bn_relu(x.to(dev), 32, (3, 3), (1, 1), "blk", training=True)
    w = bb.vs.get("blk/conv2d/kernel", (3, 3, 64, 32), "glorot").detach().cpu().to(torch.bfloat16).float()
    c = T.conv2d_same(x.float(), w, None, stride=1).to(torch.bfloat16).float()
    ref, mean, var = T.batch_norm_train(c, torch.ones(32), torch.zeros(32), 1e-5)
    ref = torch.relu(ref)
    assert (y.float().cpu() - ref).abs().max().item() <= 2.0 ** -6 * ref.abs().max().item() + 2e-3
    assert torch.allclose(bb.vs.buffer("blk/bn/moving_mean", (32,), 0.0).cpu(), mean * (1 - 0.997), atol=1e-4)
    yi = bb.conv_bn(x.to(dev), 32, (3, 3), (1, 1), "blk", training=False)          # inference path uses the moving averages
    assert yi.shape == y.shape and torch.isfinite(yi.float()).all()
    z = bb.bn_relu(x.to(dev), "pre", training=True)
    zr, _, _ = T.batch_norm_train(x.float(), torch.ones(64), torch.zeros(64), 1e-5)
    assert (z.float().cpu() - torch.relu(zr)).abs().max().item() <= 2.0 ** -7 * zr.abs().max().item() + 1e-3
