"""Data-parallel semantics of the trainer on the real kernels (contiguous split, gradients of loss_rank / N summed over
ranks, bucketed all-reduce overlapped with backward): 2 ranks sharing the one GPU (DANHIP_DP_TRANSPORT=gloo: RCCL refuses two ranks on
one device) must reach the same parameters as one process on the whole batch — except that the loss normalisation is per shard, as in
the reference (tf_replicate_model_fn averages tower losses that are each normalised by their own positives).  The RCCL data plane
(include/danhip.h danhip_comm_*, called directly by the library: no ProcessGroupNCCL) is covered with a one-rank communicator.

The file sorts LAST on purpose: these tests start child processes (rendezvous, RCCL), i.e. infrastructure that can fail for reasons that
are not numerical — under `pytest -x` they must not stand in front of the oracle comparisons."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "ddp", "worker.py")


def _free_port():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def _load(out):
    """The worker's result; the file (a copy or two of a model's parameters) is deleted at once: 21 tests x several children otherwise leave
    gigabytes in pytest's temporary directories until the session ends (a third back-to-back run of this file filled a box's /tmp)."""
    d = torch.load(out)
    try:
        os.remove(out)
    except OSError:
        pass
    return d


def _child(cmd, env, what):
    """One child, once: a crashed child fails the test (head AND tail of its stderr: `terminate called ...` is at the head)."""
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (what, r.returncode, r.stdout[-1500:], r.stderr[:3000], r.stderr[-3000:])


def _run(world, out, **extra):
    env = dict(os.environ, DANHIP_DP_TRANSPORT="gloo", MASTER_ADDR="127.0.0.1", **extra)
    if world == 1:
        cmd = [sys.executable, WORKER, out]
    else:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), WORKER, out]
    _child(cmd, env, "world %d" % world)
    return _load(out)


from tests_ddp_paths import FAKE_RCCL          # tests/ddp/libfake_rccl.so (built on demand)


def _run_fake_rccl(world, out, **extra):
    """WORLD ranks on the one GPU with transport "rccl" through tests/ddp/fake_rccl.cpp (built by __graft_entry__.build / dan_amd.build):
    everything above the thirteen nccl* symbols is the product's own code with real peers."""
    assert os.path.exists(FAKE_RCCL), "tests/ddp/libfake_rccl.so is missing: run `python -m dan_amd.build`"
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", DANHIP_RCCL_PATH=FAKE_RCCL, DANHIP_DP_NO_FALLBACK="1", DANHIP_COMM_TIMEOUT_S="240",
               FAKE_RCCL_TIMEOUT_S="200", DDP_EXPECT_RCCL_RANKS=str(world), **extra)
    env.pop("DANHIP_DP_TRANSPORT", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), WORKER, out]
    _child(cmd, env, "fake-rccl world %d" % world)
    return _load(out)


def _forced_rccl_env(**extra):
    env = dict(os.environ, DANHIP_FORCE_DIST="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", **extra)
    env.pop("DANHIP_DP_TRANSPORT", None)
    return env


def test_two_ranks_follow_the_same_trajectory_twice_and_differ_from_one_rank_only_by_shard_normalisation(dev, tmp_path):
    a = _run(2, str(tmp_path / "w2a.pt"))
    b = _run(2, str(tmp_path / "w2b.pt"))
    one = _run(1, str(tmp_path / "w1.pt"))
    scale = a["w"].abs().max().item()
    # determinism of the DP path (bucket order, all-reduce, 1/N scaling): two launches agree to fp32-atomics noise
    assert (a["w"] - b["w"]).abs().max().item() <= 1e-4 * scale
    # the all-reduced gradient is NOT the single-process gradient in general (per-shard normalisers), but it is close and finite
    assert torch.isfinite(a["w"]).all() and torch.isfinite(a["g"]).all()
    rel = (a["w"] - one["w"]).abs().max().item() / scale
    assert rel < 5e-3, rel
    assert a["g"].abs().max().item() > 0


@pytest.mark.parametrize("comm,wire,tol", [("allreduce", "f32", 1e-4), ("rs_ag", "f32", 1e-4), ("allreduce", "bf16", 2e-3), ("rs_ag", "bf16", 2e-3)])
def test_rccl_code_path_with_a_one_rank_group(comm, wire, tol, dev, tmp_path):
    """The only RCCL coverage a one-GPU box allows: DANHIP_FORCE_DIST=1 makes the process create a 1-rank RCCL communicator
    (danhip_comm_create) and run the bucketed all-reduce on its communication stream, with the weight-gradient stream beside it — same
    parameters as the plain process.
    Also the two alternative wire forms (trainer.GradBuckets): reduce-scatter + all-gather per bucket, and bf16 buckets (each gradient
    rounded to bf16 once: 2^-9 relative per element, the parameters after three lr = 1e-4 steps stay within 2e-3 of their scale)."""
    env = _forced_rccl_env(DANHIP_DP_COMM=comm, DANHIP_DP_BUCKET_DTYPE=wire)
    out = str(tmp_path / "rccl1.pt")
    _child([sys.executable, WORKER, out], env, "forced one-rank RCCL communicator")
    a = _load(out)
    one = _run(1, str(tmp_path / "plain.pt"))
    scale = one["w"].abs().max().item()
    assert (a["w"] - one["w"]).abs().max().item() <= tol * scale
    assert torch.isfinite(a["g"]).all() and a["g"].abs().max().item() > 0
    if wire == "bf16":                                 # the reduced gradient is bf16-representable: it really travelled as bf16
        assert torch.equal(a["g"], a["g"].to(torch.bfloat16).float())


@pytest.mark.parametrize("model", ["sfd", "pb", "dan", "dan_deform"])
def test_two_rank_step_equals_the_oracle_dp_step_over_shard_gradients(model, dev, tmp_path):
    """SURVEY a35 against oracle.train.dp_step (tf_replicate_model_fn.py:297-343, 458-498, 615-645): the gradient buffer a 2-rank
    run holds after its bucketed, overlapped all-reduce equals add_n over the towers of grad(loss_shard / N), each tower computed by
    ONE plain process from the same parameters — for all four graphs, with DANHIP_DP_CHECK=1 (no gradient may be written after its
    bucket was reduced).  Tolerance 1e-4 of the gradient's max-norm (fp32 atomics order in the weight-gradient kernels); the weights
    after the step equal the Momentum update of the aggregated gradient (oracle.train.momentum_sgd_step semantics: x2 on biases,
    L2 term once) at 1e-5."""
    from oracle import train as OT
    dp = _run(2, str(tmp_path / "dp.pt"), DDP_MODEL=model, DANHIP_DP_CHECK="1")
    sh = _run(1, str(tmp_path / "sh.pt"), DDP_MODEL=model, DDP_MODE="shards", DDP_SHARDS="2")
    assert torch.equal(dp["w0"], sh["w0"])

    def tower(i, loss_scale):
        assert loss_scale == 0.5                       # the shard gradients were produced with world = 2
        return sh["loss"][i], {"flat": sh["g"][i]}

    agg, reported = OT.dp_step(tower, [0, 1])
    g = agg["flat"]
    scale = g.abs().max().item()
    assert scale > 0 and torch.isfinite(dp["g1"]).all()
    err = (dp["g1"] - g).abs().max().item()
    assert err <= 1e-4 * scale, (model, err, scale)
    assert abs(dp["loss1"] - sh["loss"][0]) <= 1e-3 * abs(sh["loss"][0]) + 1e-4          # rank 0 reports its own tower's loss
    assert dp["buckets"] >= 2
    # Momentum step 0 (v = 0): w1 = w0 - lr * mult * (g + wd * w0), lr = 1e-3 * 0.1 (train_sfd.py:429-447)
    seg = sh["seg"].tolist()
    mult = torch.ones_like(g)
    wd = torch.zeros_like(g)
    for k in range(len(seg) - 1):
        mult[seg[k]:seg[k + 1]] = sh["gmult"][k]
        wd[seg[k]:seg[k + 1]] = sh["wdc"][k]
    want = sh["w0"] - 1e-4 * mult * (dp["g1"] + wd * sh["w0"])
    assert torch.allclose(dp["w1"], want, rtol=1e-5, atol=1e-7), (dp["w1"] - want).abs().max().item()


def _oracle_dp(sh):
    from oracle import train as OT

    def tower(i, loss_scale):
        assert loss_scale == 0.5
        return sh["loss"][i], {"flat": sh["g"][i]}

    return OT.dp_step(tower, [0, 1])[0]["flat"]


@pytest.mark.parametrize("comm,wire,tol", [("allreduce", "f32", 1e-4), ("rs_ag", "f32", 1e-4), ("allreduce", "bf16", 1.2e-2), ("rs_ag", "bf16", 1.2e-2)])
def test_rccl_transport_with_two_real_ranks_equals_the_oracle_dp_step(comm, wire, tol, dev, tmp_path):
    """VERDICT r5 missing 1: the RCCL data plane with MORE THAN ONE rank.  Two ranks on this one GPU, transport "rccl" bound to the
    shared-memory stand-in (danhip_comm_load(path)): the gradient buffer after the bucketed, overlapped exchange — plain all-reduce, or
    reduce-scatter + all-gather with the w = 2 shard offsets of GradBuckets._reduce, fp32 or bf16 on the wire — equals
    oracle.train.dp_step (tf_replicate_model_fn.py:297-343, 633-645: add_n of grad(loss_shard / N)) over shard gradients one plain
    process computed; DANHIP_DP_CHECK=1 (no gradient written after its bucket left).  bf16 wire: each rank's gradient is rounded once
    before the sum and the sum once after it (2^-9 each; bound 3 * 2^-8 of the max-norm)."""
    dp = _run_fake_rccl(2, str(tmp_path / "dp.pt"), DANHIP_DP_CHECK="1", DANHIP_DP_COMM=comm, DANHIP_DP_BUCKET_DTYPE=wire)
    sh = _run(1, str(tmp_path / "sh.pt"), DDP_MODE="shards", DDP_SHARDS="2")
    assert torch.equal(dp["w0"], sh["w0"])
    g = _oracle_dp(sh)
    scale = g.abs().max().item()
    err = (dp["g1"] - g).abs().max().item()
    assert scale > 0 and torch.isfinite(dp["g1"]).all() and err <= tol * scale, (comm, wire, err, scale)
    assert dp["buckets"] >= 2
    if wire == "bf16":
        assert torch.equal(dp["g1"], dp["g1"].to(torch.bfloat16).float())


@pytest.mark.parametrize("model", ["pb", "dan", "dan_deform"])
def test_rccl_transport_with_two_real_ranks_other_graphs(model, dev, tmp_path):
    dp = _run_fake_rccl(2, str(tmp_path / "dp.pt"), DDP_MODEL=model, DANHIP_DP_CHECK="1")
    sh = _run(1, str(tmp_path / "sh.pt"), DDP_MODEL=model, DDP_MODE="shards", DDP_SHARDS="2")
    g = _oracle_dp(sh)
    scale = g.abs().max().item()
    assert (dp["g1"] - g).abs().max().item() <= 1e-4 * scale and dp["buckets"] >= 2


def test_two_real_ranks_replay_the_captured_data_parallel_step(dev, tmp_path):
    """The hipGraph-captured data-parallel step WITH real peers: every rank replays [backward || exchange of finished buckets] ->
    optimizer as one graph (the stand-in's collectives are memcpy + host nodes on the buckets' stream, capturable like RCCL's kernels);
    after warm-up + 2 replays the parameters equal the eager 2-rank run's after 3 steps."""
    a = _run_fake_rccl(2, str(tmp_path / "eager.pt"))
    b = _run_fake_rccl(2, str(tmp_path / "graph.pt"), DDP_MODE="graph")
    assert a["step"] == b["step"] == 3 and b["buckets"] >= 2
    scale = a["w"].abs().max().item()
    assert (a["w"] - b["w"]).abs().max().item() <= 1e-4 * scale
    assert torch.isfinite(b["g"]).all() and b["g"].abs().max().item() > 0


def test_a_rank_that_dies_mid_job_ends_its_peer_within_the_deadline(dev, tmp_path):
    """without ProcessGroupNCCL nothing watched the collectives.  Rank 1 exits after its first step (DDP_DIE_RANK);
    rank 0's next exchange waits for a peer that is gone — trainer.CommWatch (ncclCommGetAsyncError poll + a deadline on the step's
    event) must end rank 0 with DEADLINE_EXIT_CODE and a message instead of hanging in the stream."""
    import time
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2", DANHIP_RCCL_PATH=FAKE_RCCL,
               DANHIP_DP_NO_FALLBACK="1", DANHIP_COMM_TIMEOUT_S="20", FAKE_RCCL_TIMEOUT_S="8", DDP_DIE_RANK="1", DDP_STEPS="6")
    env.pop("DANHIP_DP_TRANSPORT", None)
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, WORKER, str(tmp_path / ("die%d.pt" % r))], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    took = time.time() - t0
    assert procs[1].returncode == 17, outs[1]                                   # the rank that "died"
    assert procs[0].returncode == 75, (procs[0].returncode, outs[0][1][-2000:])  # trainer.DEADLINE_EXIT_CODE
    assert "FATAL" in outs[0][1] and ("RCCL error" in outs[0][1] or "did not complete" in outs[0][1]), outs[0][1][-2000:]
    assert took < 200, took


def test_data_parallel_step_replayed_as_one_hipgraph(dev, tmp_path):
    """DetectorTrainer.enable_graph on a data-parallel trainer: forward, backward, the bucketed RCCL all-reduce on its side stream and the
    fused optimizer captured as ONE hipGraph (the buckets' stream forks from / joins the capturing stream through events; the
    collectives are the library's own RCCL calls, recorded like kernels).  The only RCCL communicator a one-GPU box allows has one rank;
    its captured run must land on the eager forced-rank run's parameters after the same three steps (fp32-atomics noise), with more than
    one bucket in flight.  The worker ends in the prescribed order: graph dropped, device drained, communicator destroyed."""
    outs = {}
    for mode in ("dp", "graph"):
        out = str(tmp_path / (mode + ".pt"))
        _child([sys.executable, WORKER, out], _forced_rccl_env(DDP_MODE=mode), mode)
        outs[mode] = _load(out)
    a, b = outs["dp"], outs["graph"]
    assert a["step"] == b["step"] == 3 and b["buckets"] >= 2
    scale = a["w"].abs().max().item()
    assert (a["w"] - b["w"]).abs().max().item() <= 1e-4 * scale
    assert torch.isfinite(b["g"]).all() and b["g"].abs().max().item() > 0


def test_comm_entry_points_on_a_one_rank_communicator(dev, tmp_path):
    """include/danhip.h danhip_comm_*: unique id -> communicator -> all-reduce / reduce-scatter / all-gather in every wire dtype on a
    one-rank communicator are the identity (sum over one rank), asynchronous on the caller's stream; info / version answer; bad
    arguments return DANHIP_EINVAL with a message instead of reaching RCCL.  Child process: RCCL's helper threads stay out of pytest."""
    code = r"""
import ctypes, sys, torch
sys.path.insert(0, %r)
from dan_amd import _lib
from dan_amd.trainer import RcclComm
torch.cuda.set_device(0)
c = RcclComm(0, 1, 0)
assert c.version >= 20000, c.version
n, r, d = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
_lib.call("danhip_comm_info", c.handle, ctypes.byref(n), ctypes.byref(r), ctypes.byref(d))
assert (n.value, r.value, d.value) == (1, 0, 0)
side = torch.cuda.Stream()
for dt in (torch.float32, torch.bfloat16, torch.float16):
    x = torch.randn(1 << 20, device="cuda").to(dt)
    want = x.clone()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        c.all_reduce(x)
        y = torch.empty_like(x)
        c.reduce_scatter(y, x)
        z = torch.empty_like(x)
        c.all_gather(z, y)
    torch.cuda.current_stream().wait_stream(side)
    assert torch.equal(x, want) and torch.equal(y, want) and torch.equal(z, want), dt
L = _lib.lib()
bad = L.danhip_comm_allreduce_sum(c.handle, _lib.ptr(x), 16, 7, _lib.stream())
assert bad == -1 and b"dtype" in L.danhip_last_error()
assert L.danhip_comm_allreduce_sum(None, _lib.ptr(x), 16, 0, _lib.stream()) == -1
assert L.danhip_comm_create(None, 1, 0, 0, None) == -1
c.close()
c.close()                      # idempotent
print("COMM_OK")
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "COMM_OK" in r.stdout, (r.returncode, r.stdout[-1500:], r.stderr[:3000], r.stderr[-3000:])


def test_bench_line_of_a_two_rank_run_on_one_gpu(dev):
    """bench.py as the driver launches it for N > 1 (`python bench.py --gpus 2` spawns torch.distributed.run): two ranks share the one GPU of
    this box with the host-staged data plane (DANHIP_DP_TRANSPORT=gloo; RCCL refuses two ranks on one device) - what is exercised is the
    multi-rank plumbing of the line itself: gloo control plane, barrier + max-over-ranks timing, the weak `value` and the strong-scaling leg
    (global batch 8 on 2 ranks = 4 images per rank as two 2-image towers... here one 4-image tower), the roofline legs, orderly shutdown."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(DANHIP_DP_TRANSPORT="gloo", DANHIP_BENCH_RCCL_LOG="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch-per-gpu", "2", "--size", "128",
           "--strong-global-batch", "8", "--repeats", "2", "--eager", "--no-cpu-baseline", "--no-eval"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[:3000], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 4 and d["value"] > 0
    assert d["repeats"]["regions"] == 2 and len(d["repeats"]["ms_per_step"]) == 2
    s = d["strong"]
    assert s["global_batch"] == 8 and s["n_gpus"] == 2 and s["batch_per_gpu"] == 4 and s["towers_per_gpu"] * s["tower_batch"] == 4 and s["value"] > 0
    s16 = d["strong16"]                                # SURVEY 8(e)'s series: global batch 16 on 2 ranks = 8 images per rank, eager
    assert s16["global_batch"] == 16 and s16["n_gpus"] == 2 and s16["batch_per_gpu"] == 8 and s16["value"] > 0
    assert d["roofline"]["bound"] == "mfma" and d["roofline"]["frac"] > 0
    assert d["config"]["dp_transport"] == "gloo" and d["config"]["rccl_ranks"] == 1


def test_bench_line_of_a_two_rank_run_on_the_rccl_transport(dev):
    """The line the driver asks for at N > 1, on the transport it will use: `python bench.py --gpus 2` with the data plane on the library's own
    danhip_comm_* calls (bound to the shared-memory stand-in, so that two ranks fit on this one GPU): unique-id broadcast, ncclCommInitRank(2)
    under its deadline, bucketed all-reduce overlapped with backward, CommWatch ticking every step, the weak `value`, both strong legs
    (`strong`: global batch 8 as towers; `strong16`: 16 / 2 = 8 images per rank) and the orderly shutdown (communicator before process group)."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "DANHIP_DP_TRANSPORT")}
    env.update(DANHIP_RCCL_PATH=FAKE_RCCL, DANHIP_DP_NO_FALLBACK="1", DANHIP_BENCH_RCCL_LOG="0", DANHIP_COMM_TIMEOUT_S="240", FAKE_RCCL_TIMEOUT_S="200")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch-per-gpu", "2", "--size", "128",
           "--strong-global-batch", "8", "--repeats", "2", "--eager", "--no-cpu-baseline", "--no-eval"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.returncode, r.stdout[-1500:], r.stderr[:3000], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["global_batch"] == 4 and d["value"] > 0
    assert d["config"]["rccl_ranks"] == 2 and "rccl 29999" in d["config"]["dp_transport"], d["config"]
    assert d["strong"]["n_gpus"] == 2 and d["strong"]["value"] > 0 and d["strong16"]["batch_per_gpu"] == 8 and d["strong16"]["value"] > 0
    assert d["roofline"]["bound"] == "mfma" and d["roofline"]["frac"] > 0
