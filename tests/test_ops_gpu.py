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
