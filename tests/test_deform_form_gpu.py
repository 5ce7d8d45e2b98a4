"""The deformable backward chooses its form on the device from the offsets' statistic (csrc/deform_conv.hip: gather form for small offsets,
fp32-atomics scatter form when most taps have a corner outside the gather window): both forms must give the same gradients, and the
automatic choice must be one of them."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("sigma", [0.4, 3.0])
def test_gather_and_scatter_forms_of_the_deformable_backward_agree(sigma, dev):
    from dan_amd import _lib, ops
    N, H, W, C, dg = 2, 40, 36, 128, 2
    g = torch.Generator().manual_seed(7)
    x = torch.randn((N, H, W, C), generator=g).to(ops.ACT).to(dev)
    off = (torch.randn((N, H, W, dg * 18), generator=g) * sigma).to(ops.ACT).to(dev)
    dS = torch.randn((N * H * W, 9 * C), generator=g).to(ops.ACT).to(dev)

    def run(form):
        _lib.lib().danhip_set_option(b"deform_bwd_form", form)
        dx, doff = torch.empty_like(x), torch.empty_like(off)
        ws = torch.empty((x.numel() + 64,), dtype=torch.float32, device=dev)
        _lib.call("danhip_deform_sample_bwd", _lib.ptr(x), _lib.ptr(off), _lib.ptr(dS), _lib.ptr(dx), _lib.ptr(doff), N, H, W, C, 3, 3, 1, 1, dg, 0,
                  _lib.ptr(ws), _lib.stream())
        torch.cuda.synchronize()
        return dx.float(), doff.float(), int(ws[-64:].view(torch.int32)[0].item())

    try:
        gx, go, far = run(1)
        sx, so, _ = run(2)
        ax, ao, far_a = run(0)
    finally:
        _lib.lib().danhip_set_option(b"deform_bwd_form", 0)
    pairs = N * H * W * dg * 9
    assert far == far_a and ((far > pairs // 20 * 3) == (sigma > 1.0)), (far, pairs)
    # dX: fp32 sums in different orders, rounded once to 16 bits; dOffset: the same 64-channel reductions
    scale = gx.abs().max().item()
    assert (gx - sx).abs().max().item() <= 2e-2 * scale
    assert (go - so).abs().max().item() <= 2e-2 * go.abs().max().item()
    want_x, want_o = (sx, so) if sigma > 1.0 else (gx, go)
    assert torch.equal(ao, want_o) and (ax - want_x).abs().max().item() <= 2e-2 * scale
