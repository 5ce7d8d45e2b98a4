"""Test-time pipeline (eval_dan.py:95-297) on the GPU against oracle/evalpipe.py: the image resize is bit-exact, box voting
has exact cluster membership / scores and float32-rounded float64 means, and the whole multi-scale pipeline — driven by a
deterministic stand-in for the network so both sides see identical raw detections — matches detection for detection."""
import numpy as np
import pytest
import torch

from oracle import evalpipe as E

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("h,w,f", [(37, 53, 0.5), (64, 48, 0.25), (101, 77, 0.75), (33, 65, 1.25), (40, 40, 1.5), (29, 31, 1.75), (50, 70, 2.0),
                                   (123, 211, 0.3137), (17, 19, 3.7), (2, 2, 5.0), (480, 640, 0.69)])
def test_resize_bit_exact(h, w, f, dev):
    from dan_amd import eval_dan as P
    rng = np.random.RandomState(h * 1000 + w)
    img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
    ref = E.cv2_resize_linear_u8(img, f, f)
    got = P.resize_image(torch.from_numpy(img).to(dev), f, f).cpu().numpy()
    assert got.shape == ref.shape
    assert np.array_equal(got, ref)


def _clustered_dets(seed, n_clusters, per, n_single, ties=False):
    rng = np.random.RandomState(seed)
    rows = []
    for _ in range(n_clusters):
        cx, cy, s = rng.rand() * 900 + 50, rng.rand() * 600 + 50, rng.rand() * 80 + 8
        k = rng.randint(1, per + 1)
        j = rng.randn(k, 4) * s * 0.08
        sc = rng.rand(k)
        rows.append(np.stack([cx - s / 2 + j[:, 0], cy - s / 2 + j[:, 1], cx + s / 2 + j[:, 2], cy + s / 2 + j[:, 3], sc], 1))
    if n_single:
        c = rng.rand(n_single, 2) * 5000 + 2000
        rows.append(np.concatenate([c, c + 10, rng.rand(n_single, 1)], 1))
    det = np.concatenate(rows).astype(np.float32).astype(np.float64)         # float32-representable values in a float64 array
    if ties:
        det[:, 4] = np.round(det[:, 4] * 16) / 16
    det[rng.rand(det.shape[0]) < 0.02, 2] -= 500                              # a few inverted (degenerate) boxes
    return det[rng.permutation(det.shape[0])]


@pytest.mark.parametrize("seed,nc,per,ns,ties", [(0, 5, 4, 3, False), (1, 60, 12, 40, False), (2, 300, 20, 500, True), (3, 1, 1, 0, False),
                                                 (4, 900, 6, 100, False)])
def test_bbox_vote_matches_oracle(seed, nc, per, ns, ties, dev):
    from dan_amd import eval_dan as P
    det = _clustered_dets(seed, nc, per, ns, ties)
    ref = E.bbox_vote(det)
    got = P.bbox_vote(torch.from_numpy(det).to(dev)).cpu().numpy()
    assert got.shape == ref.shape and got.dtype == np.float32
    assert np.array_equal(got[:, 4], ref[:, 4])                               # cluster order, membership (max score) exact
    assert np.allclose(got[:, :4], ref[:, :4], rtol=2e-7, atol=0), np.abs(got - ref).max()   # float64 means, summation order differs


def test_bbox_vote_batch_and_empty(dev):
    from dan_amd import eval_dan as P
    dets = [_clustered_dets(10, 40, 8, 10), np.zeros((0, 5)), _clustered_dets(11, 3, 3, 50)]
    outs = P.bbox_vote_batch([torch.from_numpy(d).to(dev) for d in dets])
    for d, o in zip(dets, outs):
        ref = E.bbox_vote(d) if d.shape[0] else np.zeros((0, 5), np.float32)
        assert o.shape == ref.shape
        assert np.array_equal(o.cpu().numpy()[:, 4], ref[:, 4])
    big = _clustered_dets(12, 1200, 5, 0)                                     # more clusters than max_per_image: truncation
    ref = E.bbox_vote(big, max_per_image=100)
    got = P.bbox_vote(torch.from_numpy(big).to(dev), max_per_image=100).cpu().numpy()
    assert got.shape == ref.shape == (100, 5) and np.array_equal(got[:, 4], ref[:, 4])


def _fake_net_np(image):
    """Deterministic stand-in for sess.run: boxes / scores derived from the image content with integer arithmetic only."""
    h, w = image.shape[:2]
    key = (int(image.astype(np.int64).sum()) + 7919 * h + 104729 * w) % (2 ** 31 - 1)
    rng = np.random.RandomState(key)
    n = 1500
    cy, cx = rng.rand(n) * h, rng.rand(n) * w
    s = np.exp(rng.rand(n) * np.log(40)) * 6
    face = rng.randint(0, 12, n)                                              # 12 "faces" get many hits -> clusters across scales
    fy, fx, fs = (np.arange(12) * 37 % 11 + 1) / 12.0 * h, (np.arange(12) * 53 % 11 + 1) / 12.0 * w, (np.arange(12) % 4 + 1) * 0.06 * min(h, w)
    hit = rng.rand(n) < 0.5
    cy = np.where(hit, fy[face] + rng.randn(n) * fs[face] * 0.05, cy)
    cx = np.where(hit, fx[face] + rng.randn(n) * fs[face] * 0.05, cx)
    s = np.where(hit, fs[face] * (1 + rng.randn(n) * 0.05), s)
    boxes = np.stack([cy - s / 2, cx - s / 2, cy + s / 2, cx + s / 2], 1).astype(np.float32)
    scores = np.where(hit, 0.5 + rng.rand(n) * 0.5, rng.rand(n) * 0.3).astype(np.float32)
    return boxes, scores


def _fake_net_torch(image):
    b, s = _fake_net_np(image.cpu().numpy())
    return torch.from_numpy(b).to(image.device), torch.from_numpy(s).to(image.device)


@pytest.mark.parametrize("h,w", [(240, 320), (683, 1024), (1400, 2000)])
def test_pipeline_matches_oracle_with_identical_raw_detections(h, w, dev):
    from dan_amd import eval_dan as P
    rng = np.random.RandomState(h + w)
    img = rng.randint(0, 256, (h, w, 3)).astype(np.uint8)
    timg = torch.from_numpy(img).to(dev)
    shrink, max_shrink = E.get_shrink(h, w)
    max_shrink = min(max_shrink, 2.6)                                         # bound the enlarged copies (host RAM of the oracle side)
    for fn in ("detect_face", "flip_test"):
        ref = getattr(E, fn)(_fake_net_np, img, shrink)
        got = getattr(P, fn)(_fake_net_torch, timg, shrink).cpu().numpy()
        assert got.dtype == ref.dtype and np.array_equal(got, ref), fn
    refs = list(E.multi_scale_test(_fake_net_np, img, max_shrink)) + [E.multi_scale_test_pyramid(_fake_net_np, img, max_shrink)]
    gots = list(P.multi_scale_test(_fake_net_torch, timg, max_shrink)) + [P.multi_scale_test_pyramid(_fake_net_torch, timg, max_shrink)]
    for r, g in zip(refs, gots):
        assert np.array_equal(g.cpu().numpy(), r)
    alld = np.vstack([E.detect_face(_fake_net_np, img, shrink), E.flip_test(_fake_net_np, img, shrink)] + refs)
    ref = E.bbox_vote(alld)
    got = P.bbox_vote(torch.from_numpy(alld).to(dev)).cpu().numpy()
    assert got.shape == ref.shape and ref.shape[0] > 5
    assert np.array_equal(got[:, 4], ref[:, 4]) and np.allclose(got[:, :4], ref[:, :4], rtol=2e-7, atol=0)


@pytest.mark.parametrize("model", ["sfd", "dan"])
def test_detect_image_real_network_odd_size(model, dev):
    """The whole loop body of eval_*.py:452-459 on the real graphs at a size that is not a multiple of anything
    (ragged tiles in every conv / pool / resize kernel): finite, sorted into the vote's order, inside the score range."""
    from dan_amd import eval_dan as P, eval_sfd as PS
    from dan_amd import train_dan, train_sfd
    rng = np.random.RandomState(5)
    img = torch.from_numpy(rng.randint(0, 256, (203, 331, 3)).astype(np.uint8)).to(dev)
    if model == "sfd":
        net = P.Detector(train_sfd.SFDModel(device=dev), lambda h, w, d: train_sfd.AnchorConfig(h, w, d))
        dets = PS.detect_image(net, img)
    else:
        net = P.Detector(train_dan.DANModel(device=dev), train_dan.dan_anchor_config)
        dets = P.detect_image(net, img)
    d = dets.cpu().numpy()
    assert d.ndim == 2 and d.shape[1] == 5 and d.shape[0] <= 750
    assert np.isfinite(d).all() and (d[:, 4] >= 0).all() and (d[:, 4] <= 1).all()
    # single-scale consistency: the detector run on a resized copy equals detect_face's own resize + run
    half = P.resize_image(img, 0.5, 0.5)
    b, s = net(half)
    got = P.detect_face(net, img, 0.5)
    top = min(s.shape[0] - 1, 1125)
    assert got.shape == (top, 5)
    assert torch.equal(got[:, 4], torch.sort(s, descending=True).values[:top])
    k = int(torch.argmax(s))
    if int((s == s.max()).sum()) == 1:
        assert torch.equal(got[0, :4], torch.stack((b[k, 1], b[k, 0], b[k, 3], b[k, 2])) / torch.tensor(0.5, device=dev))


class _FakeBatchNet(object):
    """The deterministic stand-in as a `net` with the Detector's two entry points: per image, or B images of one size in one call."""

    def __call__(self, image):
        return _fake_net_torch(image)

    def batch(self, images):
        outs = [_fake_net_torch(images[b]) for b in range(images.shape[0])]
        return torch.stack([o[0] for o in outs]), torch.stack([o[1] for o in outs])


@pytest.mark.parametrize("h,w,pyramid", [(240, 320, True), (400, 300, False), (683, 1024, True)])
def test_batched_pipeline_equals_the_per_image_pipeline(h, w, pyramid, dev):
    """detect_images (all scales of B images without a host round trip, size filters as validity masks, ONE voting launch
    with device-side counts) against detect_image per image, bit for bit, with identical raw detections on both sides."""
    from dan_amd import eval_dan as P
    rng = np.random.RandomState(h * 3 + w)
    imgs = torch.from_numpy(rng.randint(0, 256, (3, h, w, 3)).astype(np.uint8)).to(dev)
    net = _FakeBatchNet()
    out, num = P.detect_images(net, imgs, pyramid=pyramid)
    assert out.shape == (3, P.MAX_PER_IMAGE, 5) and num.dtype == torch.int32
    for b in range(3):
        ref = P.detect_image(net, imgs[b], pyramid=pyramid)
        n = int(num[b].item())
        assert n == ref.shape[0] and n > 5
        assert torch.equal(out[b, :n], ref)


def test_batched_pipeline_real_network(dev):
    """The same on the real S3FD graph: a batch of ONE runs the same forward shapes as the serial path (bit-identical); a batch of three
    gives finite, score-ordered detections for every image."""
    from dan_amd import eval_dan as P
    from dan_amd import train_sfd
    rng = np.random.RandomState(9)
    imgs = torch.from_numpy(rng.randint(0, 256, (3, 203, 331, 3)).astype(np.uint8)).to(dev)
    net = P.Detector(train_sfd.SFDModel(device=dev), lambda h, w, d: train_sfd.AnchorConfig(h, w, d))
    out1, num1 = P.detect_images(net, imgs[:1], pyramid=False)
    ref = P.detect_image(net, imgs[0], pyramid=False)
    assert int(num1[0].item()) == ref.shape[0] and torch.equal(out1[0, :ref.shape[0]], ref)
    out, num = P.detect_images(net, imgs, pyramid=False)
    for b in range(3):
        d = out[b, :int(num[b].item())].cpu().numpy()
        assert d.shape[0] > 0 and np.isfinite(d).all() and (d[:, 4] >= 0).all() and (d[:, 4] <= 1).all()
