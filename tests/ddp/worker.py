"""Rank body of tests/test_ddp_gpu.py.

mode "dp" (default): WORLD_SIZE ranks (all on cuda:0, DANHIP_DP_TRANSPORT=gloo; or one forced RCCL rank) train data-parallel on contiguous shards
of one global batch; rank 0 saves the all-reduced gradient and the parameters after step 1 and after step 2.
mode "graph": as "dp", with the step captured as one hipGraph (bucketed all-reduce included; RCCL groups only) after one eager warm-up step.
mode "shards" (DDP_MODE=shards, one plain process): the same global batch cut into DDP_SHARDS contiguous shards; for each shard the
gradient of loss_shard / N is computed from the SAME initial parameters (what one tower of tf_replicate_model_fn.py:297-302 computes) and
saved, for the oracle's dp_step to aggregate.
DDP_MODEL selects the graph: sfd | pb | dan | dan_deform.
With DANHIP_RCCL_PATH=tests/ddp/libfake_rccl.so (tests/ddp/fake_rccl.cpp: the nccl* entry points over shared memory, duplicate devices
accepted) the SAME code runs the transport "rccl" with WORLD_SIZE > 1 ranks on one GPU: unique-id broadcast, ncclCommInitRank(N), the
bucketed all-reduce / reduce-scatter + all-gather with real peers, eager or captured."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch

from dan_amd import ops, synthetic
from dan_amd.trainer import init_distributed, shutdown_distributed

out = sys.argv[1]
mode = os.environ.get("DDP_MODE", "dp")
which = os.environ.get("DDP_MODEL", "sfd")
rank, world, local = init_distributed()
dev = torch.device("cuda", local)
GB, S = 4, 128                                            # global batch
imgs = synthetic.make_images(GB, S, S, dev, seed=31)
gts = synthetic.make_gt_boxes(GB, S, S, seed=32, max_faces=5)


def build(world_):
    """-> (trainer, per-image target tensors as a tuple / dict, fn(slice) -> train_step arguments)"""
    if which == "sfd":
        from dan_amd.train_sfd import AnchorConfig, SFDModel, SFDTrainer
        anchors = AnchorConfig(S, S, dev)
        loc_t, cls_t, _ = anchors.encode_batch(gts)
        tr = SFDTrainer(SFDModel(device=dev, seed=9), world=world_)
        return tr, lambda sl: (imgs[sl].contiguous(), loc_t[sl].contiguous(), cls_t[sl].contiguous())
    if which == "pb":
        from dan_amd.train_pb import PBAnchorTargets, PBModel, PBTrainer
        tg = PBAnchorTargets(S, S, dev)
        t = tg.encode_batch(gts)
        tr = PBTrainer(PBModel(device=dev, seed=9), world=world_)
        return tr, lambda sl: (imgs[sl].contiguous(), {k: tuple(a[sl].contiguous() for a in v) for k, v in t.items()})
    from dan_amd.train_dan import DANModel, DANTrainer, dan_anchor_config, encode_batch_dan
    anchors = dan_anchor_config(S, S, dev)
    loc_t, cls_t, mgt = encode_batch_dan(anchors, gts)
    tr = DANTrainer(DANModel(device=dev, seed=9, deform=(which == "dan_deform")), anchors, world=world_)
    return tr, lambda sl: (imgs[sl].contiguous(), loc_t[sl].contiguous(), cls_t[sl].contiguous(), mgt[sl].contiguous())


if mode == "shards":
    n = int(os.environ.get("DDP_SHARDS", "2"))
    tr, args_of = build(n)                                # world = n: every loss term's gradient carries 1 / n
    assert not tr.buckets.enabled
    w0 = tr.flat.w.clone()
    per = GB // n
    gs, losses = [], []
    for i in range(n):
        tr.flat.w.copy_(w0)
        tr.flat.v.zero_()
        ops.WEIGHT_EPOCH += 1
        ops.repack_all()
        tr.step_no = 0
        if hasattr(tr, "_routing_ctr"):
            tr._routing_ctr.zero_()                       # every rank starts its routing stream at the same counter
        tr.train_step(*args_of(slice(i * per, (i + 1) * per)))
        torch.cuda.synchronize()
        gs.append(tr.flat.g.clone().cpu())
        losses.append(tr.loss_values()["total"] - tr.loss_values()["l2"])
    torch.save({"w0": w0.cpu(), "g": gs, "loss": losses, "seg": tr.flat.seg.cpu(), "gmult": tr.flat.gmult.cpu(), "wdc": tr.flat.wdc.cpu(),
                "names": tr.flat.names}, out)
    sys.exit(0)

per = GB // world
sl = slice(rank * per, (rank + 1) * per)                 # contiguous split (tf_replicate_model_fn.py:458-498)
tr, args_of = build(world)
forced = os.environ.get("DANHIP_FORCE_DIST") == "1"
assert tr.buckets.enabled == (world > 1 or forced)
if forced:
    assert tr.buckets.device_collectives and tr.buckets.rccl is not None    # the library's own RCCL communicator: the weight-gradient stream stays on beside the buckets
if os.environ.get("DDP_EXPECT_RCCL_RANKS"):                                 # the stand-in library: a REAL multi-rank communicator behind danhip_comm_*
    assert tr.buckets.transport == "rccl" and tr.buckets.rccl.world == int(os.environ["DDP_EXPECT_RCCL_RANKS"]) == world, (tr.buckets.transport, world)
    assert tr.buckets.rccl.version == 29999 and tr.buckets.watch is not None
w0 = tr.flat.w.clone()
if mode == "graph":                                      # the data-parallel step (bucketed all-reduce included) replayed as ONE hipGraph
    tr.enable_graph(*args_of(sl), warmup=1)
    for _ in range(2):
        tr.train_step(*args_of(sl))
    torch.cuda.synchronize()
    if rank == 0:
        torch.save({"w": tr.flat.w.cpu(), "g": tr.flat.g.cpu(), "step": tr.step_no, "buckets": len(tr.buckets.bounds)}, out)
    shutdown_distributed(tr)                             # graph first, then the communicator, then the process group
    sys.exit(0)
tr.train_step(*args_of(sl))
torch.cuda.synchronize()
if os.environ.get("DDP_DIE_RANK"):                      # failure-detection test: one rank leaves without a word, the other keeps stepping
    if rank == int(os.environ["DDP_DIE_RANK"]):
        os._exit(17)
    for _ in range(int(os.environ.get("DDP_STEPS", "6"))):
        tr.train_step(*args_of(sl))
    torch.cuda.synchronize()                             # (never reached: CommWatch ends the process first)
    sys.exit(1)
g1, w1 = tr.flat.g.clone(), tr.flat.w.clone()
loss1 = tr.loss_values()["total"] - tr.loss_values()["l2"]
tr.train_step(*args_of(sl))
tr.train_step(*args_of(sl))                           # (a third step: the graph mode's warm-up + 2 replays = 3 steps)
torch.cuda.synchronize()
if rank == 0:
    torch.save({"w": tr.flat.w.cpu(), "g": tr.flat.g.cpu(), "w0": w0.cpu(), "g1": g1.cpu(), "w1": w1.cpu(), "loss1": loss1,
                "buckets": len(tr.buckets.bounds), "step": tr.step_no}, out)
shutdown_distributed(tr)
