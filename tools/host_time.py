import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dan_amd import synthetic
from dan_amd.train_sfd import AnchorConfig, SFDModel, SFDTrainer
dev = torch.device("cuda:0")
B, S = 16, 640
imgs = synthetic.make_images(B, S, S, dev, seed=1)
gts = synthetic.make_gt_boxes(B, S, S, seed=2)
anchors = AnchorConfig(S, S, dev)
loc_t, cls_t, _ = anchors.encode_batch(gts)
tr = SFDTrainer(SFDModel(device=dev), world=1)
for _ in range(3):
    tr.train_step(imgs, loc_t, cls_t)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    tr.train_step(imgs, loc_t, cls_t)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("host issue %.2f ms, until GPU done %.2f ms" % ((t1 - t0) * 1e3, (t2 - t0) * 1e3))
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    tr.train_step(imgs, loc_t, cls_t)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
