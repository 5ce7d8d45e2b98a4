"""Rank body of tests/test_ddp_gpu.py: WORLD_SIZE ranks (all on cuda:0, gloo backend) train S3FD data-parallel on contiguous shards of
one global batch; rank 0 saves the parameters after 2 steps for comparison with the single-process run on the whole batch."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from dan_amd import synthetic
from dan_amd.train_sfd import AnchorConfig, SFDModel, SFDTrainer
from dan_amd.trainer import init_distributed

out = sys.argv[1]
rank, world, local = init_distributed()
dev = torch.device("cuda", local)
GB, S = 4, 128                                            # global batch
imgs = synthetic.make_images(GB, S, S, dev, seed=31)
gts = synthetic.make_gt_boxes(GB, S, S, seed=32, max_faces=5)
anchors = AnchorConfig(S, S, dev)
loc_t, cls_t, _ = anchors.encode_batch(gts)
per = GB // world
sl = slice(rank * per, (rank + 1) * per)                 # contiguous split (tf_replicate_model_fn.py:458-498)
tr = SFDTrainer(SFDModel(device=dev, seed=9), world=world)
forced = os.environ.get("DANHIP_FORCE_DIST") == "1"
assert tr.buckets.enabled == (world > 1 or forced)
if forced:
    assert tr.buckets.device_collectives                 # RCCL group: the weight-gradient stream stays on beside the buckets
for _ in range(2):
    tr.train_step(imgs[sl].contiguous(), loc_t[sl].contiguous(), cls_t[sl].contiguous())
torch.cuda.synchronize()
if rank == 0:
    torch.save({"w": tr.flat.w.cpu(), "g": tr.flat.g.cpu()}, out)
if torch.distributed.is_initialized():
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
