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
                  _lib.ptr(ws), ws.numel() * 4, _lib.stream())
        torch.cuda.synchronize()
        return dx.float(), doff.float(), int(ws[-64:].view(torch.int32)[0].item())

    try:
        gx, go, far = run(1)              # gather, +-2 px window
        sx, so, _ = run(2)                # fp32-atomics scatter
        nx, no, _ = run(3)                # gather, +-1 px window (everything beyond it through the far-corner atomics)
        ax, ao, far_a = run(0)            # chosen on the device
    finally:
        _lib.lib().danhip_set_option(b"deform_bwd_form", 0)
    pairs = N * H * W * dg * 9
    assert far == far_a and ((far > pairs // 20 * 3) == (sigma > 1.0)), (far, pairs)
    # dX: fp32 sums in different orders, rounded once to 16 bits; dOffset: the same 64-channel reductions
    scale = gx.abs().max().item()
    for other_x, other_o in ((sx, so), (nx, no)):
        assert (gx - other_x).abs().max().item() <= 2e-2 * scale
        assert (go - other_o).abs().max().item() <= 2e-2 * go.abs().max().item()
    want_x, want_o = (sx, so) if sigma > 1.0 else (gx, go)           # sigma = 0.4: 1.2 % of the pairs leave [-1, 1) -> the +-2 window
    assert torch.equal(ao, want_o) and (ax - want_x).abs().max().item() <= 2e-2 * scale


def test_small_offsets_take_the_narrow_gather_window(dev):
    """Offsets inside [-1, 1) (the zero-initialised offset convolution of the reference and the first training steps): the +-1 window."""
    from dan_amd import _lib, ops
    N, H, W, C, dg = 1, 24, 28, 64, 1
    g = torch.Generator().manual_seed(3)
    x = torch.randn((N, H, W, C), generator=g).to(ops.ACT).to(dev)
    off = ((torch.rand((N, H, W, dg * 18), generator=g) - 0.5) * 1.9).to(ops.ACT).to(dev)
    dS = torch.randn((N * H * W, 9 * C), generator=g).to(ops.ACT).to(dev)
    outs = []
    try:
        for form in (0, 3, 1):
            _lib.lib().danhip_set_option(b"deform_bwd_form", form)
            dx, doff = torch.empty_like(x), torch.empty_like(off)
            ws = torch.empty((x.numel() + 64,), dtype=torch.float32, device=dev)
            _lib.call("danhip_deform_sample_bwd", _lib.ptr(x), _lib.ptr(off), _lib.ptr(dS), _lib.ptr(dx), _lib.ptr(doff), N, H, W, C, 3, 3, 1, 1, dg, 0,
                      _lib.ptr(ws), ws.numel() * 4, _lib.stream())
            torch.cuda.synchronize()
            outs.append((dx.clone(), doff.clone(), ws[-64:].view(torch.int32)[:2].tolist()))
    finally:
        _lib.lib().danhip_set_option(b"deform_bwd_form", 0)
    assert outs[0][2] == [0, 0]                                       # nothing outside [-2, 2) nor [-1, 1)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])     # automatic = the narrow window, bit for bit
    assert (outs[0][0].float() - outs[2][0].float()).abs().max().item() <= 2e-2 * outs[2][0].float().abs().max().item()


@pytest.mark.parametrize("form,sigma", [(3, 0.3), (1, 0.7), (0, 0.0), (0, 1.2)])
@pytest.mark.parametrize("shape", [(2, 13, 21, 128, 2), (1, 40, 36, 64, 1), (3, 5, 5, 256, 4)])
def test_tile_kernels_of_the_gather_forms_against_the_per_item_kernels(form, sigma, shape, dev):
    """Round 5: both gather windows run as LDS-staged tile kernels (8 x 16 tiles; ragged edges, tiles larger than the map, an accumulated dX);
    the per-item / wave-per-pixel kernels they replace stay behind the option `deform_dx_untiled` and must agree: dX sums the same candidates
    in the same order, dOffset reduces the same 64-channel products in another lane order."""
    from dan_amd import _lib, ops
    N, H, W, C, dg = shape
    g = torch.Generator().manual_seed(11)
    x = torch.randn((N, H, W, C), generator=g).to(ops.ACT).to(dev)
    off = (torch.randn((N, H, W, dg * 18), generator=g) * sigma).to(ops.ACT).to(dev)
    dS = torch.randn((N * H * W, 9 * C), generator=g).to(ops.ACT).to(dev)
    dx0 = torch.randn((N, H, W, C), generator=g).to(ops.ACT).to(dev)
    outs = []
    try:
        _lib.lib().danhip_set_option(b"deform_bwd_form", form)
        for untiled in (0, 1):
            _lib.lib().danhip_set_option(b"deform_dx_untiled", untiled)
            dx, doff = dx0.clone(), torch.empty_like(off)
            ws = torch.full((x.numel() + 64,), float("nan"), dtype=torch.float32, device=dev)       # the call zeroes what it reads
            _lib.call("danhip_deform_sample_bwd", _lib.ptr(x), _lib.ptr(off), _lib.ptr(dS), _lib.ptr(dx), _lib.ptr(doff), N, H, W, C, 3, 3, 1, 1, dg, 1,
                      _lib.ptr(ws), ws.numel() * 4, _lib.stream())
            torch.cuda.synchronize()
            outs.append((dx.float(), doff.float()))
    finally:
        _lib.lib().danhip_set_option(b"deform_bwd_form", 0)
        _lib.lib().danhip_set_option(b"deform_dx_untiled", 0)
    (tx, to), (ux, uo) = outs
    assert torch.isfinite(tx).all() and torch.isfinite(to).all()
    assert (tx - ux).abs().max().item() <= 1e-2 * ux.abs().max().item()          # (bf16 roundings of nearly equal fp32 sums)
    assert (to - uo).abs().max().item() <= 1e-2 * max(uo.abs().max().item(), 1e-6)
