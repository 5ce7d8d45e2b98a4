"""Host issue time against GPU time of one eager training step: is the step launch-bound?  usage: python tools/host_time.py [sfd|pb|dan|dan_deform] [profile]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dan_amd import synthetic

which = sys.argv[1] if len(sys.argv) > 1 else "sfd"
dev = torch.device("cuda:0")
B, S = 16, 640
imgs = synthetic.make_images(B, S, S, dev, seed=1)
gts = synthetic.make_gt_boxes(B, S, S, seed=2)
if which == "sfd":
    from dan_amd.train_sfd import AnchorConfig, SFDModel, SFDTrainer
    anchors = AnchorConfig(S, S, dev)
    loc_t, cls_t, _ = anchors.encode_batch(gts)
    tr = SFDTrainer(SFDModel(device=dev), world=1)
    args = (imgs, loc_t, cls_t)
elif which == "pb":
    from dan_amd.train_pb import PBAnchorTargets, PBModel, PBTrainer
    tr = PBTrainer(PBModel(device=dev), world=1)
    args = (imgs, PBAnchorTargets(S, S, dev).encode_batch(gts))
else:
    from dan_amd.train_dan import DANModel, DANTrainer, dan_anchor_config, encode_batch_dan
    anchors = dan_anchor_config(S, S, dev)
    tr = DANTrainer(DANModel(device=dev, deform=which == "dan_deform"), anchors, world=1)
    args = (imgs,) + encode_batch_dan(anchors, gts)
for _ in range(3):
    tr.train_step(*args)
torch.cuda.synchronize()
for rep in range(4):
    t0 = time.perf_counter()
    tr.train_step(*args)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s: host issue %.2f ms, until GPU done %.2f ms" % (which, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
if len(sys.argv) > 2:
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        tr.train_step(*args)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(25)
