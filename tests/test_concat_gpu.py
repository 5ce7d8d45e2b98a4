"""tf.concat(axis=-1) / residual-add backward (danhip_slice_deliver; ops.concat / ops.add) and the ReLU-masked average-pool backward:
the kernel against numpy bit for bit, and the direct gradient hand-off against plain autograd through the same layer kernels
(ops.USE_SLOTS = False) on a small context-module-shaped graph (net/danet.py:842-918: branches -> concat -> 1x1 -> + input)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bf(t):
    return t.to(torch.bfloat16)


@pytest.mark.parametrize("ldy,c0,C,masked,acc", [(256, 64, 32, True, 0), (256, 0, 64, False, 1), (256, 85, 171, True, 0), (256, 0, 85, True, 1),
                                                 (24, 8, 8, True, 1), (16, 3, 5, False, 0)])
def test_slice_deliver_kernel(ldy, c0, C, masked, acc, dev):
    from dan_amd._lib import call, ptr, stream
    M = 333
    Cpad = (C + 7) // 8 * 8
    g = torch.Generator().manual_seed(ldy + c0 + C)
    dy = _bf(torch.randn(M, ldy, generator=g))
    mask = _bf(torch.randn(M, C, generator=g))
    old = _bf(torch.randn(M, Cpad, generator=g))
    want = dy[:, c0:c0 + C].float()
    if masked:
        want = torch.where(mask.float() > 0, want, torch.zeros(()))
    want = torch.cat([want, torch.zeros(M, Cpad - C)], 1)
    if acc:
        want = want + old.float()
    want = _bf(want)
    dyd, md, out = dy.to(dev), mask.to(dev), old.to(dev).clone()
    call("danhip_slice_deliver", ptr(dyd), ldy, c0, C, ptr(md) if masked else None, C, ptr(out), Cpad, acc, M, stream())
    torch.cuda.synchronize()
    assert torch.equal(out.cpu().view(torch.int16), want.view(torch.int16))


def test_slice_deliver_rejects_bad_slices(dev):
    from dan_amd._lib import DanhipError, call, ptr, stream
    a = torch.zeros(4, 16, dtype=torch.bfloat16, device=dev)
    with pytest.raises(DanhipError):
        call("danhip_slice_deliver", ptr(a), 16, 12, 8, None, 8, ptr(a), 8, 0, 4, stream())          # slice past the row
    with pytest.raises(DanhipError):
        call("danhip_slice_deliver", ptr(a), 16, 0, 8, None, 8, ptr(a), 12, 0, 4, stream())          # Cpad not a multiple of 8


def _graph(ops, x, ws):
    """branches (ReLU 16, ragged ReLU 11, avg-pool -> ReLU 8, linear 8 -> 5 channels of padding) -> concat -> 1x1 ReLU -> + x."""
    a = ops.conv2d(x, ws["a"], ws["ab"], relu=True)
    b = ops.conv2d(x, ws["b"], ws["bb"], relu=True)
    c = ops.conv2d(ops.avg_pool_2x2_s1(x), ws["c"], ws["cb"], relu=True)
    d = ops.conv2d(a, ws["d"], ws["db"], relu=True)
    e = ops.conv2d(x, ws["e"], ws["eb"], relu=True)
    h = ops.concat([a, b, c, d, e])
    y = ops.add(ops.conv2d(h, ws["r"], ws["rb"], relu=True), x)
    return ops.conv2d(y, ws["o"], ws["ob"], relu=False) , ops.conv2d(h, ws["o2"], ws["o2b"], relu=False)


@pytest.mark.parametrize("relu_input", [False, True])
def test_concat_add_avgpool_hand_off_equals_autograd(relu_input, dev):
    from dan_amd import ops
    g = torch.Generator().manual_seed(5)
    C = 32
    shapes = {"a": (1, 1, C, 16), "b": (1, 1, C, 11), "c": (1, 1, C, 8), "d": (3, 3, 16, 8), "e": (3, 1, C, 5), "r": (1, 1, 48, C), "o": (3, 3, C, 8),
              "o2": (1, 1, 48, 8), "in": (3, 3, 8, C)}
    ws = {}
    for k, s in shapes.items():
        ws[k] = (torch.randn(s, generator=g) * (2.0 / (s[0] * s[1] * s[2])) ** 0.5).to(dev).requires_grad_(True)
        ws[k + "b"] = (0.1 * torch.randn(s[3], generator=g)).to(dev).requires_grad_(True)
    x0 = _bf(torch.randn(2, 9, 13, 8, generator=g)).to(dev)
    dy1 = _bf(torch.randn(2, 9, 13, 8, generator=g)).to(dev)
    dy2 = _bf(torch.randn(2, 9, 13, 8, generator=g)).to(dev)
    res = {}
    for slots in (True, False):
        ops.USE_SLOTS = slots
        try:
            for p in ws.values():
                p.grad = None
            x = ops.conv2d(x0, ws["in"], ws["inb"], relu=relu_input)          # the block input is itself a layer output (ReLU or not)
            y1, y2 = _graph(ops, x, ws)
            torch.autograd.backward([y1, y2], [dy1, dy2])
            torch.cuda.synchronize()
            res[slots] = ({k: p.grad.clone() for k, p in ws.items()}, y1.detach().clone())
        finally:
            ops.USE_SLOTS = True
    assert torch.equal(res[True][1], res[False][1])
    for k in ws:
        a, b = res[True][0][k], res[False][0][k]
        assert b.abs().max().item() > 0, k
        # the two routes sum the same bf16 terms in a different order: one bf16 rounding of the largest entry
        assert (a - b).norm().item() <= 2.0 ** -6 * b.norm().item() + 1e-6, (k, (a - b).norm().item(), b.norm().item())


def test_ragged_concat_feeding_another_concat(dev):
    """a concat whose width is not a multiple of 8 (16 + 11 = 27) owns a channel-padded gradient slot (32); a downstream concat
    delivers into it and the ragged concat's backward hands its inputs their slices — same gradients as the autograd route."""
    from dan_amd import ops
    g = torch.Generator().manual_seed(9)
    C = 16
    shapes = {"a": (1, 1, C, 16), "b": (1, 1, C, 11), "c": (1, 1, C, 5), "o": (3, 3, 32, 8), "in": (3, 3, 8, C)}
    ws = {}
    for k, s in shapes.items():
        ws[k] = (torch.randn(s, generator=g) * (2.0 / (s[0] * s[1] * s[2])) ** 0.5).to(dev).requires_grad_(True)
        ws[k + "b"] = (0.1 * torch.randn(s[3], generator=g)).to(dev).requires_grad_(True)
    x0 = _bf(torch.randn(2, 9, 13, 8, generator=g)).to(dev)
    dy = _bf(torch.randn(2, 9, 13, 8, generator=g)).to(dev)
    res = {}
    for slots in (True, False):
        ops.USE_SLOTS = slots
        try:
            for p in ws.values():
                p.grad = None
            x = ops.conv2d(x0, ws["in"], ws["inb"], relu=True)
            a = ops.conv2d(x, ws["a"], ws["ab"], relu=True)
            b = ops.conv2d(x, ws["b"], ws["bb"], relu=True)
            c = ops.conv2d(x, ws["c"], ws["cb"], relu=False)
            h = ops.concat([ops.concat([a, b]), c])                      # 27 (ragged) + 5 = 32
            y = ops.conv2d(h, ws["o"], ws["ob"], relu=False)
            y.backward(dy)
            torch.cuda.synchronize()
            res[slots] = ({k: p.grad.clone() for k, p in ws.items()}, y.detach().clone())
        finally:
            ops.USE_SLOTS = True
    assert torch.equal(res[True][1], res[False][1])
    for k in ws:
        a_, b_ = res[True][0][k], res[False][0][k]
        assert b_.abs().max().item() > 0, k
        assert (a_ - b_).norm().item() <= 2.0 ** -6 * b_.norm().item() + 1e-6, (k, (a_ - b_).norm().item(), b_.norm().item())
