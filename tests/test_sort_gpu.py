"""libdanhip's stable descending arg-sort (csrc/sort.hip) against torch.sort(descending=True, stable=True): the candidate ordering of
utility/bbox_util.py:61-91 (tf.nn.top_k: ties -> lower index first) and eval_dan.py:255 (argsort()[::-1]: ties -> higher index first).
Index work: bit-exact."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [1, 2, 5, 1000, 8191, 8192, 8193, 34125, 65536, 87360])
@pytest.mark.parametrize("levels", [0, 17])          # 0: distinct random floats; 17: quantised to 17 values (long runs of ties)
def test_argsort_desc_matches_torch_stable_sort(n, levels, dev):
    from dan_amd import ops
    g = torch.Generator().manual_seed(n * 31 + levels)
    s = torch.rand((n,), generator=g)
    if levels:
        s = torch.floor(s * levels) / levels
        s[::7] = -s[::7]                              # negative values and both zeros
        s[1::11] = -0.0
    s = s.to(dev)
    want = torch.sort(s, descending=True, stable=True).indices
    got = ops.argsort_desc(s)
    assert got.dtype == torch.int64 and torch.equal(got, want)
    want_hi = (n - 1) - torch.sort(s.flip(0), descending=True, stable=True).indices
    assert torch.equal(ops.argsort_desc(s, ties_high_index_first=True), want_hi)


def test_argsort_rejects_a_short_workspace_and_orders_infinities(dev):
    from dan_amd import _lib
    s = torch.tensor([0.5, float("inf"), -float("inf"), 0.5, 3.0], device=dev)
    from dan_amd import ops
    assert ops.argsort_desc(s).tolist() == [1, 4, 0, 3, 2]
    idx = torch.empty((5,), dtype=torch.int32, device=dev)
    ws = torch.empty((64,), dtype=torch.uint8, device=dev)
    with pytest.raises(_lib.DanhipError):
        _lib.call("danhip_argsort_desc_f32", _lib.ptr(s), 5, 0, _lib.ptr(idx), _lib.ptr(ws), 64, _lib.stream())


def test_nans_of_either_sign_sort_first_like_torch(dev):
    """Diverged logits: torch.sort(descending=True) on the CPU and numpy's argsort()[::-1] (eval_dan.py:255) put EVERY NaN first (stable among
    themselves); sign-bit NaNs used to land behind -inf and 0xFFFFFFFF collided with the padding key.  The reference here is the
    CPU sort: torch's own stable GPU sort on this ROCm build orders sign-bit NaNs by their bits, i.e. after -inf."""
    from dan_amd import ops
    for n in (9, 8200):
        g = torch.Generator().manual_seed(n)
        s = torch.rand((n,), generator=g) - 0.5
        import numpy as np
        a = s.numpy().view(np.uint32)
        a[0], a[n - 1] = 0xFFFFFFFF, 0xFFC00001                       # sign-bit NaNs with payloads (a Python float would lose them)
        s[3], s[4], s[n - 2] = float("nan"), float("inf"), -float("inf")
        want = torch.sort(s, descending=True, stable=True).indices    # CPU
        s = s.to(dev)
        got = ops.argsort_desc(s).cpu()
        assert torch.equal(got, want), (got[:6].tolist(), want[:6].tolist())
        assert got[:3].tolist() == [0, 3, n - 1] and got[3].item() == 4 and got[-1].item() == n - 2
        assert ops.argsort_desc(s, ties_high_index_first=True)[:3].tolist() == [n - 1, 3, 0]


def test_sorted_candidates_feed_nms_like_before(dev):
    """sort_bboxes / nms_bboxes (bbox_util.py:61-91) on scores with ties: same selections as with torch's stable sort."""
    from dan_amd.utility import bbox_util as BU
    g = torch.Generator().manual_seed(5)
    n = 3000
    sc = (torch.floor(torch.rand((n,), generator=g) * 50) / 50).to(dev)
    yx = torch.rand((n, 2), generator=g) * 300
    hw = torch.rand((n, 2), generator=g) * 60 + 4
    bb = torch.cat([yx, yx + hw], dim=1).to(dev)
    order = torch.sort(sc, descending=True, stable=True).indices
    s2, y0, x0, y1, x1 = BU.sort_bboxes(sc, bb[:, 0], bb[:, 1], bb[:, 2], bb[:, 3], 400)
    assert torch.equal(s2, sc[order][:400]) and torch.equal(y0, bb[order, 0][:400])
    ks, kb = BU.nms_bboxes(sc, bb, 200, 0.3)
    assert ks.shape[0] > 10 and bool((ks[:-1] >= ks[1:]).all())
