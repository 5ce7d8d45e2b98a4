"""Data-parallel semantics of the trainer on the real kernels (contiguous split, gradients of loss_rank / N summed over
ranks, bucketed all-reduce overlapped with backward): 2 ranks sharing the one GPU (gloo, DANHIP_DIST_BACKEND) must reach the same
parameters as one process on the whole batch — except that the loss normalisation is per shard, as in the reference
(tf_replicate_model_fn averages tower losses that are each normalised by their own positives)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "ddp", "worker.py")


def _run(world, out, port):
    env = dict(os.environ, DANHIP_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    if world == 1:
        cmd = [sys.executable, WORKER, out]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(port), WORKER, out]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    return torch.load(out)


def test_two_ranks_follow_the_same_trajectory_twice_and_differ_from_one_rank_only_by_shard_normalisation(dev, tmp_path):
    a = _run(2, str(tmp_path / "w2a.pt"), 29621)
    b = _run(2, str(tmp_path / "w2b.pt"), 29622)
    one = _run(1, str(tmp_path / "w1.pt"), 0)
    scale = a["w"].abs().max().item()
    # determinism of the DP path (bucket order, all-reduce, 1/N scaling): two launches agree to fp32-atomics noise
    assert (a["w"] - b["w"]).abs().max().item() <= 1e-4 * scale
    # the all-reduced gradient is NOT the single-process gradient in general (per-shard normalisers), but it is close and finite
    assert torch.isfinite(a["w"]).all() and torch.isfinite(a["g"]).all()
    rel = (a["w"] - one["w"]).abs().max().item() / scale
    assert rel < 5e-3, rel
    assert a["g"].abs().max().item() > 0


def test_rccl_code_path_with_a_one_rank_group(dev, tmp_path):
    """The only RCCL coverage a one-GPU box allows: DANHIP_FORCE_DIST=1 makes rank 0 of a 1-rank NCCL (= RCCL) group run the bucketed
    all-reduce on its communication stream, with the weight-gradient stream beside it — same parameters as the plain process."""
    env = dict(os.environ, DANHIP_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    env.pop("DANHIP_DIST_BACKEND", None)
    out = str(tmp_path / "rccl1.pt")
    r = subprocess.run([sys.executable, WORKER, out], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    a = torch.load(out)
    one = _run(1, str(tmp_path / "plain.pt"), 0)
    scale = one["w"].abs().max().item()
    assert (a["w"] - one["w"]).abs().max().item() <= 1e-4 * scale
    assert torch.isfinite(a["g"]).all() and a["g"].abs().max().item() > 0
