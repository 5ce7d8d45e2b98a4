"""Synthetic inputs of SURVEY 8(d): seeded uint8 images and ground-truth boxes (no dataset offline)."""
import math

import torch

SEED = 20180817  # the reference's tf_random_seed (train_sfd.py:93)


def make_images(batch, height, width, device, seed=SEED):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(0, 256, (batch, height, width, 3), generator=g, dtype=torch.uint8).to(device)


def make_gt_boxes(batch, height, width, seed=SEED, max_faces=40):
    """Per image G ~ U{1..max_faces} boxes, side ~ logU(8,300) px, centre uniform, clipped to the image, filtered
    h>6, w>3 (preprocessing/dan_preprocessing.py:726-728).  Returns a list of [G,4] fp32 (ymin,xmin,ymax,xmax) CPU tensors."""
    g = torch.Generator().manual_seed(seed + 1)
    out = []
    for _ in range(batch):
        n = int(torch.randint(1, max_faces + 1, (1,), generator=g))
        side = torch.exp(torch.rand(n, generator=g) * (math.log(300.0) - math.log(8.0)) + math.log(8.0))
        ar = 0.8 + 0.4 * torch.rand(n, generator=g)
        h, w = side * ar, side / ar
        cy = torch.rand(n, generator=g) * height
        cx = torch.rand(n, generator=g) * width
        b = torch.stack([cy - h / 2, cx - w / 2, cy + h / 2, cx + w / 2], -1)
        b[:, 0].clamp_(0, height - 1); b[:, 2].clamp_(0, height - 1)
        b[:, 1].clamp_(0, width - 1); b[:, 3].clamp_(0, width - 1)
        b = torch.round(b)
        keep = ((b[:, 2] - b[:, 0]) > 6) & ((b[:, 3] - b[:, 1]) > 3)
        b = b[keep]
        if b.shape[0] == 0:
            b = torch.tensor([[height / 4.0, width / 4.0, height / 2.0, width / 2.0]])
        out.append(b.to(torch.float32))
    return out
