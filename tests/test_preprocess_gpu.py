"""On-device training input pipeline (dan_amd/preprocessing) against the oracle restatement (oracle/preprocess.py): with the same
random draws the two produce the same crop window / boxes / flip, and the fused device pass (colour chain, mean-filled crop,
legacy bilinear resize, flip, uint8 saturation, means, BGR) reproduces the materialised numpy pipeline."""
import numpy as np
import pytest
import torch

from oracle import preprocess as O

pytestmark = pytest.mark.gpu
ANCHOR_SCALES = [16., 32., 64., 128., 256., 512.]


def _image_and_boxes(seed, h, w, n):
    rng = np.random.RandomState(seed)
    base = rng.randint(0, 256, (h // 8 + 1, w // 8 + 1, 3)).astype(np.float32)
    img = np.kron(base, np.ones((8, 8, 1), np.float32))[:h, :w] + rng.randn(h, w, 3) * 12          # blocky + noise: smooth and sharp parts
    img = np.clip(img, 0, 255).astype(np.uint8)
    cy, cx = rng.rand(n) * h, rng.rand(n) * w
    s = np.exp(rng.rand(n) * np.log(12)) * 10
    b = np.stack([np.clip(cy - s / 2, 0, h - 1), np.clip(cx - s / 2, 0, w - 1), np.clip(cy + s / 2, 0, h - 1), np.clip(cx + s / 2, 0, w - 1)], 1)
    return img, b.astype(np.float32)


@pytest.mark.parametrize("seed,h,w,n", [(0, 240, 320, 6), (1, 413, 577, 12), (2, 96, 128, 1), (3, 700, 500, 30), (4, 333, 333, 3), (5, 128, 96, 2),
                                        (6, 480, 640, 9), (7, 200, 900, 5)])
def test_preprocess_for_train_matches_oracle(seed, h, w, n, dev):
    from dan_amd.preprocessing import dan_preprocessing as P
    img, boxes = _image_and_boxes(seed, h, w, n)
    out_shape = (160, 160) if seed % 2 else (320, 320)
    ref_img, ref_boxes, info = O.preprocess_for_train(img, boxes, out_shape, ANCHOR_SCALES, O.Draws(100 + seed))
    got_img, got_boxes = P.preprocess_for_train(torch.from_numpy(img).to(dev), boxes, out_shape, ANCHOR_SCALES, draws=P.Draws(100 + seed))
    assert got_boxes.shape == ref_boxes.shape and np.array_equal(got_boxes, ref_boxes), info
    g = got_img.float().cpu().numpy()
    assert g.shape == (out_shape[0], out_shape[1], 8) and np.all(g[..., 3:] == 0)
    ref16 = torch.from_numpy(ref_img).to(got_img.dtype).float().numpy()                 # the network input is stored in 16 bits
    diff = np.abs(g[..., :3] - ref16)
    # identical integer pixel levels except where a float op order differs by an ulp right at a uint8 truncation boundary
    assert (diff > 0).mean() < 2e-3 and diff.max() <= 1.01, (info, (diff > 0).mean(), diff.max())


def test_colour_chain_each_op_alone(dev):
    from dan_amd.preprocessing import dan_preprocessing as P
    img, _ = _image_and_boxes(11, 64, 96, 1)
    timg = torch.from_numpy(img).to(dev)
    for ops in ([("brightness", 0.1)], [("saturation", 1.4)], [("saturation", 0.5)], [("hue", 0.17)], [("hue", -0.2)], [("contrast", 1.5)],
                [("contrast", 0.5), ("hue", 0.1), ("brightness", -0.05), ("saturation", 1.2)], []):
        dist, _ = O.distort_color((img.astype(np.float32) * np.float32(1.0 / 255)).astype(np.float32), ops)
        ref = O.finish(O.resize_bilinear_legacy(O.crop_with_mean_fill(dist, (0, 0, 64, 96)), 64, 96), False)
        got = P.augment_image(timg, ops, (0, 0, 64, 96), False, (64, 96)).float().cpu().numpy()[..., :3]
        ref16 = torch.from_numpy(ref).to(torch.bfloat16).float().numpy()
        d = np.abs(got - ref16)
        assert (d > 0).mean() < 2e-3 and d.max() <= 1.01, (ops, (d > 0).mean(), d.max())


def test_window_outside_the_image_is_mean_filled_and_flip(dev):
    from dan_amd.preprocessing import dan_preprocessing as P
    img, _ = _image_and_boxes(12, 50, 60, 1)
    win = (-20, -10, 100, 100)                                                   # leaves the image on every side
    dist = (img.astype(np.float32) * np.float32(1.0 / 255)).astype(np.float32)
    for flip in (False, True):
        ref = O.finish(O.resize_bilinear_legacy(O.crop_with_mean_fill(dist, win), 80, 80), flip)
        got = P.augment_image(torch.from_numpy(img).to(dev), [], win, flip, (80, 80)).float().cpu().numpy()[..., :3]
        ref16 = torch.from_numpy(ref).to(torch.bfloat16).float().numpy()
        assert np.abs(got - ref16).max() <= 1.01 and (np.abs(got - ref16) > 0).mean() < 2e-3
    # far outside: exactly the mean colour -> (B, G, R) - means = trunc(mean*255.5/255) - mean
    far = P.augment_image(torch.from_numpy(img).to(dev), [], (500, 500, 16, 16), False, (8, 8)).float().cpu().numpy()
    exp = np.trunc(np.asarray([103.94, 116.78, 123.68], np.float32) / np.float32(255.) * np.float32(255.5)) - np.asarray([103.94, 116.78, 123.68], np.float32)
    assert np.allclose(far[0, 0, :3], torch.tensor(exp).to(torch.bfloat16).float().numpy())


def test_augmented_image_feeds_the_network(dev):
    from dan_amd.preprocessing import dan_preprocessing as P
    from dan_amd.train_sfd import SFDModel
    img, boxes = _image_and_boxes(13, 300, 420, 8)
    x, b = P.preprocess_for_train(torch.from_numpy(img).to(dev), boxes, (128, 128), ANCHOR_SCALES, draws=P.Draws(5))
    model = SFDModel(device=dev)
    with torch.no_grad():
        feats = model.backbone.get_featmaps(x.unsqueeze(0), training=False)
    assert feats[0].shape[1:3] == (32, 32) and all(torch.isfinite(f.float()).all() for f in feats)
