"""Debug: phase timing of the data-parallel step with the weight-gradient stream on/off (run under torch.distributed.run, gloo, 2 ranks
sharing one GPU).  Synchronises at phase boundaries, so the numbers are for diagnosis only."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dan_amd import ops, synthetic
from dan_amd.train_sfd import AnchorConfig, SFDModel, SFDTrainer
from dan_amd.trainer import init_distributed, lr_schedule

rank, world, local = init_distributed()
dev = torch.device("cuda", local)
B, S = 16, 640
imgs = synthetic.make_images(B, S, S, dev, seed=1 + rank)
gts = synthetic.make_gt_boxes(B, S, S, seed=2 + rank)
anchors = AnchorConfig(S, S, dev)
loc_t, cls_t, _ = anchors.encode_batch(gts)
tr = SFDTrainer(SFDModel(device=dev), world=world)
for _ in range(2):
    tr.train_step(imgs, loc_t, cls_t)
torch.cuda.synchronize()
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for it in range(3):
    t0 = T()
    tr.flat.zero_grad(); tr.buckets.begin_step()
    ops.GRAD_READY_HOOK = tr._hook if tr.buckets.enabled else None
    terms = tr.loss_terms(imgs, loc_t, cls_t)
    t1 = T()
    accs = [t[2] for t in terms]
    ops.wgrad_overlap_begin()
    h0 = time.perf_counter()
    torch.autograd.backward(accs, [torch.full_like(a, tr.loss_scale) for a in accs])
    h1 = time.perf_counter()
    ops.GRAD_READY_HOOK = None
    t2 = T()
    tr.buckets.finish()
    t3 = T()
    ops.wgrad_overlap_join()
    tr.flat.sgd_step(1e-4, 0.9)
    t4 = T()
    if rank == 0:
        print("overlap=%s  fwd %.1f  bwd host-issue %.1f, until done %.1f  finish %.1f  join+sgd %.1f ms" % (os.environ.get("DANHIP_WGRAD_STREAM", "1"), (t1 - t0) * 1e3, (h1 - h0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3), flush=True)
if world > 1:
    torch.distributed.barrier(); torch.distributed.destroy_process_group()
