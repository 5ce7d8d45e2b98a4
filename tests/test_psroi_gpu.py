"""Deformable PS-ROI pooling (cpp/Deform/deform_psroi_pooling_op_gpu.cu) through custom_op.deform_psroi_pool against the numpy
restatement (oracle/deform.py): forward bit-exact (values and sample counts), backward to fp32-atomics accuracy; ROIs that leave the
map, degenerate ROIs, several classes / parts / samples, and the no_trans form."""
import numpy as np
import pytest
import torch

from oracle import deform as OD

pytestmark = pytest.mark.gpu


def _case(seed, B, H, W, output_dim, G, P, part, spp, ncls, no_trans, scale, trans_std):
    rng = np.random.RandomState(seed)
    C = output_dim * G * G
    data = rng.randn(B, C, H, W).astype(np.float32)
    R = 7
    x1 = rng.rand(R) * W / scale * 0.8 - 4; y1 = rng.rand(R) * H / scale * 0.8 - 4
    bw = rng.rand(R) * W / scale * 0.7 + 0.2; bh = rng.rand(R) * H / scale * 0.7 + 0.2
    rois = np.stack([rng.randint(0, B, R).astype(np.float32), x1, y1, x1 + bw, y1 + bh], 1).astype(np.float32)
    rois[0, 1:] = [3.4, 2.6, 3.4, 2.6]                                  # degenerate: a single point
    rois[1, 1:] = [-30, -30, -20, -20]                                  # completely outside: every sample skipped -> count 0
    trans = (rng.randn(R, 2 * ncls, max(part, 1), max(part, 1)) * 0.5).astype(np.float32)
    at = dict(spatial_scale=scale, output_dim=output_dim, group_size=G, pooled_size=P, part_size=part, sample_per_part=spp, trans_std=trans_std,
              no_trans=no_trans)
    return data, rois, trans, at


@pytest.mark.parametrize("cfg", [(0, 2, 12, 15, 4, 3, 3, 3, 2, 2, False, 0.25, 0.1), (1, 1, 9, 9, 2, 2, 4, 2, 3, 1, False, 0.5, 0.2),
                                 (2, 2, 10, 8, 3, 1, 2, 2, 1, 3, False, 1.0, 0.05), (3, 1, 7, 11, 2, 3, 3, 3, 2, 1, True, 0.125, 0.0),
                                 # more channels per class than lanes in a wave; fewer part cells than bins (several bins share one shift)
                                 (4, 1, 8, 8, 70, 1, 2, 2, 2, 1, False, 0.5, 0.1), (5, 2, 14, 14, 6, 2, 6, 3, 2, 2, False, 0.25, 0.3)])
def test_deform_psroi_pool_forward_backward(cfg, dev):
    from dan_amd.utility import custom_op
    data, rois, trans, at = _case(*cfg)
    top_ref, cnt_ref = OD.deform_psroi_pool_forward(data, rois, trans, **at)
    d = torch.from_numpy(data).to(dev).requires_grad_(True)
    t = torch.from_numpy(trans).to(dev).requires_grad_(True)
    top, cnt = custom_op.deform_psroi_pool(d, torch.from_numpy(rois).to(dev), t, at["spatial_scale"], at["output_dim"], at["group_size"],
                                           at["pooled_size"], at["part_size"], at["sample_per_part"], at["trans_std"], at["no_trans"])
    assert np.array_equal(cnt.detach().cpu().numpy(), cnt_ref)
    assert np.array_equal(top.detach().cpu().numpy(), top_ref)                      # bit-exact
    assert (cnt_ref[1] == 0).all() and (top_ref[1] == 0).all()
    rng = np.random.RandomState(99)
    g = rng.randn(*top_ref.shape).astype(np.float32)
    top.backward(torch.from_numpy(g).to(dev))
    dd_ref, dt_ref = OD.deform_psroi_pool_backward(data, rois, trans, cnt_ref, g, **at)
    assert np.allclose(d.grad.cpu().numpy(), dd_ref, rtol=1e-5, atol=1e-5 * np.abs(dd_ref).max())
    if at["no_trans"]:
        assert t.grad is None
    else:
        assert np.allclose(t.grad.cpu().numpy(), dt_ref, rtol=1e-4, atol=1e-5 * max(np.abs(dt_ref).max(), 1e-3))


def test_shift_gradient_is_written_once_per_cell_without_atomics(dev):
    """The kernel owns one wave per (roi, class, part cell): the shift gradient is a plain store of a fixed-order sum — identical bits on
    every run, and no dependence on what the output buffer held before (it is not zero-filled)."""
    from dan_amd._lib import call, ptr, stream
    data, rois, trans, at = _case(5, 2, 14, 14, 6, 2, 6, 3, 2, 2, False, 0.25, 0.3)
    _, cnt_ref = OD.deform_psroi_pool_forward(data, rois, trans, **at)
    g = np.random.RandomState(3).randn(*cnt_ref.shape).astype(np.float32)
    d, r, t = (torch.from_numpy(a).to(dev) for a in (data, rois, trans))
    outs = []
    for fill in (0.0, 123.0, float("nan")):
        dd = torch.empty_like(d)
        dt = torch.full_like(t, fill)
        call("danhip_deform_psroi_pool_bwd", ptr(torch.from_numpy(g).to(dev)), ptr(torch.from_numpy(cnt_ref).to(dev)), ptr(d), ptr(r), ptr(t), ptr(dd),
             ptr(dt), 2, r.shape[0], d.shape[1], 14, 14, 6, 2, 6, 3, 2, 0.25, 0.3, 0, 2, stream())
        torch.cuda.synchronize()
        outs.append(dt.cpu())
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    assert outs[0].abs().max().item() > 0


def test_deform_psroi_pool_argument_checks(dev):
    from dan_amd._lib import DanhipError
    from dan_amd.utility import custom_op
    data = torch.zeros((1, 8, 4, 4), device=dev)
    rois = torch.zeros((1, 5), device=dev)
    trans = torch.zeros((1, 2, 2, 2), device=dev)
    with pytest.raises(ValueError):
        custom_op.deform_psroi_pool(data[0], rois, trans, 1.0, 2, 2, 2, 2)
    with pytest.raises(DanhipError):                                              # 8 channels < output_dim * group_size^2 = 18
        custom_op.deform_psroi_pool(data, rois, trans, 1.0, 2, 3, 2, 2)
