#!/usr/bin/env python3
"""Headline benchmark: S3FD (VGG-16 backbone + 6 detection heads) 640x640 bf16 TRAINING throughput
(forward + backward + gradient all-reduce + momentum-SGD step) in images/second, whole job.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)
    (DANHIP_DTYPE=fp16 python bench.py ... runs the fp16 build of the library; the default and the headline are bf16)

One "step" = one optimisation step on a synthetic batch of 16 images per GPU (BASELINE.json configs[1]); inputs
(uint8 images, encoded anchor targets) are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.
Regions, in order: W warm-up steps | the TIMED region (exactly K steps between barrier + synchronize pairs: `value`, `ms_per_step`) | two
more regions of the same K steps (`repeats`: min / median over the three, so a 2-3 % change is told from noise) | the roofline region (the
same K steps with two HIP events per convolution launch: `roofline`, `kernels`, `event_recording`) | two steps with the weight-gradient
stream off (`roofline.serialized`) | the strong-scaling leg (`strong`: BASELINE.json configs[2]'s global batch of 128 on these N GPUs,
16-image towers) | inference / target-encoder / CPU-baseline legs.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # dense bf16 MFMA peak (MI355X_MICROARCH.md, chip-level parameters)


def spawn_ranks(n):
    """`python bench.py --gpus N` typed without a launcher: start `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD
    process (never an exec, and before this process has imported torch or touched the GPU), forward its output and exit with its code.
    The children see WORLD_SIZE and take the normal path."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def rccl_debug_setup(rank):
    """Multi-rank runs describe their own collectives: before the process group exists, point RCCL's INFO log (INIT + TUNING subsystems:
    communicator shape and the algorithm / protocol chosen per collective size) at a per-rank file.  Defaults only - a caller's own
    NCCL_DEBUG* settings win.  Returns the file rank 0 parses after the run (rccl_debug_parse), or None."""
    import tempfile
    if os.environ.get("DANHIP_BENCH_RCCL_LOG", "1") == "0":
        return None
    if "NCCL_DEBUG_FILE" in os.environ:
        return os.environ["NCCL_DEBUG_FILE"].replace("%h", "host").replace("%p", str(os.getpid()))
    path = os.path.join(tempfile.gettempdir(), "danhip_rccl_%d_%d.log" % (os.getpid(), rank))
    # (the GPU boxes export NCCL_DEBUG=VERSION: anything quieter than INFO is raised - the log goes to the file, not to the terminal)
    if os.environ.get("NCCL_DEBUG", "").upper() not in ("INFO", "TRACE"):
        os.environ["NCCL_DEBUG"] = "INFO"
    os.environ.setdefault("NCCL_DEBUG_SUBSYS", "INIT,TUNING,GRAPH")
    os.environ["NCCL_DEBUG_FILE"] = path
    return path


def rccl_debug_parse(text):
    """-> {"nranks", "nnodes", "channels", "algo": {...}, "proto": {...}, "transport": [...], "version"} from an RCCL / NCCL INFO log; every field
    is best effort (None / empty when the log has no such line: e.g. a one-rank group never tunes a collective)."""
    import re
    out = {"version": None, "nranks": None, "nnodes": None, "channels": None, "algo": {}, "proto": {}, "transport": []}
    m = re.search(r"(?:RCCL|NCCL) version\s*:?\s*([0-9][^\s]*)", text)
    if m:
        out["version"] = m.group(1)
    m = re.search(r"nranks (\d+)", text) or re.search(r"nRanks 0*(\d+)", text)
    if m:
        out["nranks"] = int(m.group(1))
    m = re.search(r"nNodes (\d+)", text) or re.search(r"nnodes (\d+)", text, re.I)
    if m:
        out["nnodes"] = int(m.group(1))
    m = re.search(r"(\d+) coll channels", text) or re.search(r"Channel \d+/(\d+)", text)
    if m:
        out["channels"] = int(m.group(1))
    algo_names = {"0": "Tree", "1": "Ring", "2": "CollNetDirect", "3": "CollNetChain", "4": "NVLS", "5": "NVLSTree"}
    proto_names = {"0": "LL", "1": "LL128", "2": "Simple"}
    # "AllReduce: 33554432 Bytes -> Algo 1 proto 2 time 123.4"  (TUNING subsystem)
    for coll, nbytes, al, pr in re.findall(r"(\w+): (\d+) Bytes -> Algo (\d+) proto (\d+)", text):
        key = "%s/%dMiB" % (coll, int(nbytes) >> 20) if int(nbytes) >= (1 << 20) else "%s/%dB" % (coll, int(nbytes))
        out["algo"][key] = algo_names.get(al, al)
        out["proto"][key] = proto_names.get(pr, pr)
    # other spellings of the same information ("... algorithm RING protocol LL128 ...")
    for coll, al, pr in re.findall(r"(AllReduce|ReduceScatter|AllGather|Broadcast)[^\n]*?[Aa]lgorithm[ =:]+(\w+)[^\n]*?[Pp]rotocol[ =:]+(\w+)", text):
        out["algo"].setdefault(coll, al)
        out["proto"].setdefault(coll, pr)
    out["rings"] = len(set(re.findall(r"Ring (\d+) :", text))) or None
    out["log_lines"] = text.count("\n")
    for t in re.findall(r"via ((?:P2P|SHM|NET)[/\w]*)", text):
        if t not in out["transport"]:
            out["transport"].append(t)
    return out


def strong_plan(global_batch, world, tower_batch=16):
    """The strong-scaling leg's shape on `world` ranks: (images per rank, towers per rank, images per tower) or None when the fixed global
    batch does not split (tf_replicate_model_fn.py:461-466: the batch must divide by the number of towers)."""
    if not global_batch or global_batch % world:
        return None
    per_rank = global_batch // world
    tb = min(tower_batch, per_rank)
    if per_rank % tb:
        return None
    return per_rank, per_rank // tb, tb


def csrc_hash():
    """sha256 over the kernel sources (the stamp tools/pmc_traffic.py writes into the PMC traffic file)."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = (glob.glob(os.path.join(ROOT, "dan_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "dan_amd", "csrc", "*.h")) +
             glob.glob(os.path.join(ROOT, "dan_amd", "csrc", "*.cpp")) + [os.path.join(ROOT, "include", "danhip.h")])
    for f in sorted(files):
        h.update(open(f, "rb").read())
    return h.hexdigest()


def cpu_baseline(seconds_budget=25.0):
    """The CPU oracle (PyTorch-CPU fp32 restatement of train_sfd.py's step; TF 1.8 is not installable offline) timed on
    the host cores on a bounded sample, as BASELINE.md section 4 plans it: batch 2 at 640x640, fwd + bwd + momentum step on all host cores
    (median step), the forward of one image (BASELINE.json configs[0]) on all cores and on ONE thread."""
    import torch
    from oracle import nets as ON
    from oracle import train as OT
    torch.manual_seed(0)
    P = ON.Params(create=True, seed=20180817)
    B = 2
    img = torch.randint(0, 256, (B, 640, 640, 3), dtype=torch.uint8)
    x = ON.preprocess_synthetic(img)
    with torch.no_grad():
        ON.sfd_forward(P, x[:, :64, :64])
    params = {n: v.clone().requires_grad_(True) for n, v in P.t.items()}
    mom = {n: torch.zeros_like(v) for n, v in params.items()}
    A = 34125
    cls_t = torch.zeros((B, A), dtype=torch.int64)
    cls_t[:, ::97] = 1
    loc_t = torch.randn((B, A, 4))

    def step():
        PO = ON.Params(params)
        loc, cls = ON.sfd_forward(PO, x)
        ce, ll, _ = OT.detection_loss(cls, loc, cls_t, loc_t)
        loss = ce + ll + OT.l2_regularizer(params)
        grads = torch.autograd.grad(loss, list(params.values()))
        with torch.no_grad():
            OT.momentum_sgd_step({n: p for n, p in params.items()}, dict(zip(params.keys(), grads)), mom, 1e-4)

    step()                                   # warm-up
    t0 = time.time()
    times = []
    while True:
        s0 = time.time()
        step()
        times.append(time.time() - s0)
        if time.time() - t0 > seconds_budget or len(times) >= 10:
            break
    n = len(times)
    med = sorted(times)[n // 2]
    # BASELINE.json configs[0]: S3FD forward on ONE 640x640 image (the reference's own CPU-runnable case)
    x1 = x[:1].contiguous()

    def forward_time(budget, most):
        with torch.no_grad():
            ON.sfd_forward(ON.Params(P.t), x1)
            ts = []
            f0 = time.time()
            while True:
                a = time.time()
                ON.sfd_forward(ON.Params(P.t), x1)
                ts.append(time.time() - a)
                if time.time() - f0 > budget or len(ts) >= most:
                    break
        return sorted(ts)[len(ts) // 2], len(ts)

    cores = torch.get_num_threads()
    fdt, nf = forward_time(6.0, 5)
    torch.set_num_threads(1)                 # "plus a 1-thread run"
    try:
        f1, n1 = forward_time(8.0, 2)
    finally:
        torch.set_num_threads(cores)
    return {"value": round(B / med, 4), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "median of %d S3FD train steps (fwd+bwd+SGD), batch %d, 640x640 fp32, oracle/nets.py on PyTorch-CPU "
                      "(CPU restatement of the reference graph: TF 1.8 unavailable offline)" % (n, B),
            "forward_1x640": {"value": round(1.0 / fdt, 4), "unit": "images/sec", "ms": round(fdt * 1e3, 1), "cores": cores,
                              "sample": "median of %d S3FD forwards of one 640x640 image (BASELINE.json configs[0]), same oracle, same cores" % nf},
            "forward_1x640_one_thread": {"value": round(1.0 / f1, 4), "unit": "images/sec", "ms": round(f1 * 1e3, 1), "cores": 1,
                                         "sample": "median of %d such forwards with torch.set_num_threads(1)" % n1}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=80, help="steps of the timed region (default 80: about 1 s at batch 16)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch-per-gpu", type=int, default=16)
    ap.add_argument("--global-batch", type=int, default=0,
                    help="strong-scaling series (SURVEY 8d): fix the GLOBAL batch; each of the N ranks takes global/N images (default: weak, "
                         "--batch-per-gpu images per rank)")
    ap.add_argument("--strong-global-batch", type=int, default=128,
                    help="global batch of the strong-scaling leg printed beside the weak line (BASELINE.json configs[2]: 128 = 16 images per GPU "
                         "at 8 GPUs); each rank takes global/N images as 16-image towers (train_step_towers); 0 = skip the leg")
    ap.add_argument("--strong16-global-batch", type=int, default=16,
                    help="SURVEY 8(e)'s own strong series: 'S3FD 16 -> 2 per rank' - a fixed global batch of 16 on the N ranks, 16 / N images per "
                         "rank as one tower (tf_replicate_model_fn.py:458-498); 0 = skip")
    ap.add_argument("--repeats", type=int, default=3, help="timed regions of K steps each (the first one is `value`; all of them are in `repeats`)")
    ap.add_argument("--size", type=int, default=640)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="capture the training step (incl. the bucketed RCCL all-reduce when N > 1) in a hipGraph; off by default")
    ap.add_argument("--eager", action="store_true", help="never capture the step (below 8 images per GPU the step is host-launch-bound and the default is "
                    "hipGraph replay: ~180 launches of 5-30 us kernels)")
    ap.add_argument("--events-steps", type=int, default=0, help="steps of the roofline (event-recording) region behind the timed one (default: K; 2 for a graph run)")
    ap.add_argument("--no-eval", action="store_true", help="skip the inference-FPS leg (the 'eval FPS' half of BASELINE.json's metric)")
    ap.add_argument("--deform-offsets", type=float, default=0.0,
                    help="dan_deform: set the (zero-initialised) offset convs' biases ~ U(-R, R) pixels so the sampling kernels run on "
                         "fractional, spread-out positions (SURVEY 8d asks for a second run at R = 2)")
    ap.add_argument("--no-serialized-roofline", action="store_true",
                    help="keep the steps outside the warm-up and the timed region to ONE: skips the two that time the dominant kernel with the "
                         "weight-gradient stream off and shortens the event-recording region to one step (use under rocprofv3: its per-kernel totals "
                         "then cover warm-up + K + 1 steps)")
    ap.add_argument("--model", default="sfd", choices=["sfd", "pb", "dan", "dan_deform"],
                    help="sfd = BASELINE.json configs[1] (the metric's single-GPU configuration); the others are the per-GPU shards of configs[2..4]")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))

    global torch, dist
    import torch
    import torch.distributed as dist
    from dan_amd import _lib, ops, synthetic
    from dan_amd.trainer import init_distributed, shutdown_distributed
    from dan_amd.train_sfd import AnchorConfig, SFDModel, SFDTrainer

    rccl_log = None
    if (int(os.environ.get("WORLD_SIZE", "1")) > 1 or os.environ.get("DANHIP_FORCE_DIST") == "1") and os.environ.get("DANHIP_BENCH_DRY") != "1":
        rccl_log = rccl_debug_setup(int(os.environ.get("RANK", "0")))         # (before the communicator is created: RCCL reads these at init)
    rank, world, local = init_distributed()
    if os.environ.get("DANHIP_BENCH_DRY") == "1":      # tests/test_abi_cpu.py: the launch plumbing alone (no GPU): ranks rendezvous, rank 0 reports
        n = dist.get_world_size() if dist.is_initialized() else 1
        if dist.is_initialized():
            t = torch.ones(1)
            dist.all_reduce(t)
            n = int(t.item())
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"dry": True, "n_gpus": world, "rccl_ranks": n, "scaling": "weak",
                              "strong_plan": strong_plan(args.strong_global_batch, world)}), flush=True)
        return
    multi = world > 1                                  # a control-plane process group exists (gloo: barriers, scalar statistics)
    if world != args.gpus:
        raise SystemExit("bench.py --gpus %d was started inside a %d-rank job (WORLD_SIZE): launch it as `python bench.py --gpus %d` (it spawns its "
                         "own ranks) or with torch.distributed.run --nproc-per-node %d" % (args.gpus, world, args.gpus, args.gpus))
    dev = torch.device("cuda", local)
    B, S = args.batch_per_gpu, args.size
    if args.global_batch:
        if args.global_batch % world:
            raise SystemExit("--global-batch must divide by the number of ranks (tf_replicate_model_fn.py:461-466)")
        B = args.global_batch // world

    # synthetic shard of this rank (contiguous split of the global batch, tf_replicate_model_fn.py:458-498)
    imgs = synthetic.make_images(B, S, S, dev, seed=synthetic.SEED + rank)
    gts = synthetic.make_gt_boxes(B, S, S, seed=synthetic.SEED + 100 * rank)
    # args_of(images, gt boxes) -> train_step arguments: the input pipeline's work (anchor encoding), not part of the step
    if args.model == "sfd":
        model = SFDModel(device=dev)
        trainer = SFDTrainer(model, world=world)
        anchors = AnchorConfig(S, S, dev)
        args_of = lambda im, gt: (im,) + tuple(anchors.encode_batch(gt)[:2])
        workload = "S3FD VGG-16 backbone + 6 detection heads"
    elif args.model == "pb":
        from dan_amd.train_pb import PBAnchorTargets, PBModel, PBTrainer
        model = PBModel(device=dev)
        trainer = PBTrainer(model, world=world)
        pbt = PBAnchorTargets(S, S, dev)
        anchors = pbt.face
        args_of = lambda im, gt: (im, pbt.encode_batch(gt))
        workload = "PyramidBox (LFPN + CPM + face/head/body heads)"
    else:
        from dan_amd.train_dan import DANModel, DANTrainer, dan_anchor_config, encode_batch_dan
        model = DANModel(device=dev, deform=args.model == "dan_deform")
        anchors = dan_anchor_config(S, S, dev)
        trainer = DANTrainer(model, anchors, world=world)
        args_of = lambda im, gt: (im,) + tuple(encode_batch_dan(anchors, gt))
        workload = "DAN-Deform (deformable context module)" if args.model == "dan_deform" else "DAN (two-stage heads, dynamic anchor routing)"
    step_args = args_of(imgs, gts)
    if args.deform_offsets > 0 and args.model == "dan_deform":
        g = torch.Generator().manual_seed(synthetic.SEED + 7)
        with torch.no_grad():
            for n, p in model.vs.named():
                if n.endswith("deform_conv/conv2d/bias"):
                    p.copy_(((torch.rand(p.shape, generator=g) * 2 - 1) * args.deform_offsets).to(p.device))
        workload += ", offset biases U(-%g, %g) px" % (args.deform_offsets, args.deform_offsets)
    torch.cuda.synchronize()

    def barrier():
        """device drained on this rank -> every rank arrived (gloo, host side) -> nothing of the next region has been issued"""
        torch.cuda.synchronize()
        if multi:
            dist.barrier()

    def max_over_ranks(*vals):
        if not multi:
            return vals
        t = torch.tensor(vals, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return tuple(float(v) for v in t.tolist())

    # the per-rank shape of a strong-scaling run is host-bound in eager mode: replay the step as a hipGraph there.  S3FD (≈ 180 launches per
    # step) stops being host-bound at 4 images per GPU (eager 4.57 ms against 4.77-4.92 replayed: profiles/r4/README.md); the other graphs
    # (≈ 1000 launches) below 8.  With more than one rank the captured step holds the library's RCCL calls (trainer.RcclComm): that form
    # has only ever run on a ONE-rank communicator (one-GPU boxes), so it is taken only when --graph asks for it.
    if B < (4 if args.model == "sfd" else 8) and not args.eager and world == 1:
        args.graph = True
    if args.graph:                                    # (data-parallel steps are captured too: RCCL collectives are device-side)
        trainer.enable_graph(*step_args)
    for _ in range(args.warmup):
        trainer.train_step(*step_args)
    barrier()
    # ---- the timed region: EXACTLY K steps between two barrier + synchronize pairs, nothing else in it.  (Rounds 1-3 also recorded two HIP
    # events per convolution launch in here for the roofline; measured in round 4, that cost 4 % of the step - 13.70 against 13.16 ms - so
    # the events moved to a second, equally long region right behind this one: VERDICT r3 item 9.)
    ops.PROFILE = ops.PROFILE_BYTES = None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        trainer.train_step(*step_args)
    barrier()
    dt = time.perf_counter() - t0
    # ---- the same region again (--repeats - 1 times): `value` stays the FIRST region's; min / median over all of them say how much of a
    # difference between two lines is noise
    region_dts = [dt]
    for _ in range(max(1, args.repeats) - 1):
        r0 = time.perf_counter()
        for _ in range(args.steps):
            trainer.train_step(*step_args)
        barrier()
        region_dts.append(time.perf_counter() - r0)
    # ---- the roofline region: the same steps again, every convolution launch bracketed by HIP events on its launch stream (eager launches
    # only: events cannot be recorded inside a replayed graph, so a graph run records them over two eager steps)
    saved_graph0 = trainer._graph
    prof_steps = args.steps if not args.graph else 2
    if args.no_serialized_roofline:                   # (profiling runs: one step is enough to name the dominant kernel)
        prof_steps = 1
    if args.events_steps:
        prof_steps = args.events_steps
    trainer._graph = None
    ops.PROFILE, ops.PROFILE_BYTES = {}, {}
    t1 = time.perf_counter()
    for _ in range(prof_steps):
        trainer.train_step(*step_args)
    barrier()
    dt_prof = time.perf_counter() - t1
    prof, ops.PROFILE = ops.PROFILE, None
    prof_bytes, ops.PROFILE_BYTES = ops.PROFILE_BYTES, None
    trainer._graph = saved_graph0
    # Weight gradients run on a second stream next to the data gradients (ops.wgrad_overlap_begin), so the event-bracketed duration of
    # a backward kernel in the timed region includes the time it shares the chip.  For reference the same kernels are also timed
    # serialised (second stream off) in two extra steps AFTER the timed region; that figure is reported beside the in-region one.
    prof_serial = None
    if ops.WGRAD_STREAM and not args.no_serialized_roofline:   # every rank: the steps carry collectives
        ops.WGRAD_STREAM = False
        saved_graph, trainer._graph = trainer._graph, None
        ops.PROFILE = {}
        for _ in range(2):
            trainer.train_step(*step_args)
        torch.cuda.synchronize()
        prof_serial, ops.PROFILE = ops.PROFILE, None
        trainer._graph = saved_graph
        ops.WGRAD_STREAM = True
    # ---- the strong-scaling leg, BEHIND the roofline regions (its 128-image inputs change the allocator's state: run in front of them it left
    # a 2x outlier on the first event-bracketed launches of one kernel) (north_star: ">= 6x strong scaling at 8 GPUs"): a FIXED global batch — BASELINE.json configs[2]'s 128, i.e.
    # 16 images per GPU at N = 8 — on these N ranks.  Each rank takes its contiguous 128 / N images as 16-image towers, the way the
    # reference places more towers than devices (tf_replicate_model_fn.py:504-560): every tower's loss carries 1 / (number of towers), the
    # backward kernels accumulate into the one flat gradient buffer, ONE bucketed all-reduce and ONE optimizer step per global batch.
    # The driver's per-N lines then hold both series: `value` (weak, 16 images per GPU) and `strong.value` (128 images whatever N is).
    strong = None
    G = args.strong_global_batch
    plan = strong_plan(G, world) if (not args.global_batch and not args.graph) else None
    if plan:
        per_rank, T, tb = plan
        if True:
            towers = []
            for t in range(T):                            # rank r owns images [r * per_rank, (r + 1) * per_rank) of the global batch
                seed_t = synthetic.SEED + 1000 + rank * T + t
                towers.append(args_of(synthetic.make_images(tb, S, S, dev, seed=seed_t), synthetic.make_gt_boxes(tb, S, S, seed=seed_t + 50000)))
            ks = max(2, min(args.steps, -(-args.steps * B // per_rank)))       # about as many images as the weak region saw
            trainer.train_step_towers(towers)
            barrier()
            s0 = time.perf_counter()
            for _ in range(ks):
                trainer.train_step_towers(towers)
            barrier()
            (sdt,) = max_over_ranks(time.perf_counter() - s0)
            strong = {"scaling": "strong", "global_batch": G, "n_gpus": world, "batch_per_gpu": per_rank, "towers_per_gpu": T, "tower_batch": tb,
                      "steps": ks, "ms_per_step": round(sdt / ks * 1e3, 3), "value": round(G * ks / sdt, 3), "unit": "images/sec",
                      "what": "fixed global batch (BASELINE.json configs[2]: 128 = 16 images per GPU at 8 GPUs): every rank runs its share as "
                              "16-image towers into one gradient buffer (tf_replicate_model_fn.py:504-560), one bucketed all-reduce + one "
                              "optimizer step per global batch; speed-up over N = value(N) / value(1) of THIS field"}
            del towers
    # ---- SURVEY 8(e)'s strong series (the `strong` leg above keeps 16-image towers, so at N = 8 it is the weak line again):
    # a FIXED global batch of 16 = BASELINE.json configs[1]'s batch, 16 / N images per rank (S3FD 16 -> 2 at N = 8), one bucketed all-reduce +
    # one optimizer step per global batch.  At N = 1 the shard IS the weak line's batch: its region is quoted, nothing is re-run.
    strong16 = None
    G16 = args.strong16_global_batch
    if G16 and not args.global_batch and G16 % world == 0:
        per16 = G16 // world
        what16 = ("fixed global batch of %d (SURVEY 8e: S3FD 16 -> 2 per rank at 8 GPUs): every rank trains on its %d-image contiguous shard, one "
                  "bucketed all-reduce + one optimizer step per global batch; speed-up over N = value(N) / value(1) of THIS field" % (G16, per16))
        if per16 == B and world == 1:
            strong16 = {"scaling": "strong", "global_batch": G16, "n_gpus": 1, "batch_per_gpu": per16, "steps": args.steps, "step_launch": "hipGraph replay" if args.graph else "eager",
                        "ms_per_step": round(dt / args.steps * 1e3, 3), "value": round(G16 * args.steps / dt, 3), "unit": "images/sec",
                        "what": what16 + " (N = 1: the timed region of `value` itself)"}
        elif trainer._graph is None:
            seed16 = synthetic.SEED + 2000 + rank
            a16 = args_of(synthetic.make_images(per16, S, S, dev, seed=seed16), synthetic.make_gt_boxes(per16, S, S, seed=seed16 + 50000))
            for _ in range(3):
                trainer.train_step(*a16)
            barrier()
            k16 = max(4, min(4 * args.steps, args.steps * B // max(1, per16)))
            s0 = time.perf_counter()
            for _ in range(k16):
                trainer.train_step(*a16)
            barrier()
            (sdt,) = max_over_ranks(time.perf_counter() - s0)
            strong16 = {"scaling": "strong", "global_batch": G16, "n_gpus": world, "batch_per_gpu": per16, "steps": k16, "step_launch": "eager",
                        "ms_per_step": round(sdt / k16 * 1e3, 3), "value": round(G16 * k16 / sdt, 3), "unit": "images/sec", "what": what16}
            del a16
    dt, dt_prof = max_over_ranks(dt, dt_prof)
    region_dts = list(max_over_ranks(*region_dts))

    # ---- inference leg ("eval FPS"): eval_sfd.py / eval_dan.py single-scale graph (forward + softmax + decode [+ routing]) on the same
    # resident images, outside the timed training region; every rank runs it, the slowest rank's time counts
    eval_out = None
    if not args.no_eval:
        def timed(fn, n):
            barrier()
            e0 = time.perf_counter()
            for _ in range(n):
                fn()
            barrier()
            (et,) = max_over_ranks(time.perf_counter() - e0)
            return et
        n_eval = max(3, args.steps)
        for _ in range(2):
            model.predict(imgs, anchors)
        et = timed(lambda: model.predict(imgs, anchors), n_eval)
        eval_out = {"value": round(world * B * n_eval / et, 2), "unit": "images/sec", "batch_per_gpu": B, "ms_per_batch": round(et / n_eval * 1e3, 3),
                    "what": "single-scale inference graph (forward + softmax + box decode%s), %dx%d" % (", anchor routing" if args.model.startswith("dan") else "", S, S)}
        # the same graph on the fp32 inference path (csrc/f32_infer.hip: the build that meets the 1e-4 box tolerance, tests/test_eval_f32_gpu.py)
        model.precision = "fp32"
        b32 = B                      # (rounds 2-4 quoted it at batch 4: 275-295 img/s, a quarter of the workgroups per launch; same kernels)
        model.predict(imgs[:b32], anchors)
        et32 = timed(lambda: model.predict(imgs[:b32], anchors), 3)
        # ... and with every convolution as a split-operand product on the fp16 MFMA (csrc/split_infer.hip; the same 1e-4 bound:
        # tests/test_eval_split_gpu.py)
        model.precision = "split"
        for _ in range(2):
            model.predict(imgs, anchors)
        n_split = max(5, min(20, args.steps))
        ets = timed(lambda: model.predict(imgs, anchors), n_split)
        eval_out["split"] = {"value": round(world * B * n_split / ets, 2), "unit": "images/sec", "batch_per_gpu": B, "ms_per_batch": round(ets / n_split * 1e3, 3),
                             "what": "same graph, fp32 maps between the ops, convolutions as hi.hi + lo.hi + hi.lo products of IEEE-half limbs on "
                                     "v_mfma_f32_16x16x32_f16 with fp32 accumulation (boxes within 1e-4 of the fp32 oracle)"}
        model.precision = "act"
        eval_out["fp32"] = {"value": round(world * b32 * 3 / et32, 2), "unit": "images/sec", "batch_per_gpu": b32, "ms_per_batch": round(et32 / 3 * 1e3, 3),
                            "what": "same graph, fp32 storage + fp32-input MFMA end to end (boxes within 1e-4 of the fp32 oracle)"}

    # ---- target-encoder leg (rank 0): anchor_encoder_fn for the batch the step consumes (the reference runs it per image in tf.data on
    # the CPU; here it is one library call per batch that a pipeline would issue on a side stream).  Not part of `value`.
    enc_out = None
    if rank == 0 and not args.no_eval:
        enc_fn = {"sfd": lambda: anchors.encode_batch(gts)}.get(args.model)
        if enc_fn is not None:
            for _ in range(3):
                enc_fn()
            torch.cuda.synchronize()
            e0 = time.perf_counter()
            for _ in range(20):
                enc_fn()
            torch.cuda.synchronize()
            et = (time.perf_counter() - e0) / 20
            enc_out = {"value": round(B / et, 1), "unit": "images/sec", "ms_per_batch": round(et * 1e3, 3), "gt_boxes_in_batch": int(sum(g.shape[0] for g in gts)),
                       "what": "IoU + small-mining match + target encode for the batch, one call (danhip_encode_anchors_batched), bit-exact index work"}

    rccl_info = None
    if rank == 0 and trainer.buckets.rccl is not None and rccl_log and os.path.exists(rccl_log):
        try:
            rccl_info = rccl_debug_parse(open(rccl_log, errors="replace").read())
            rccl_info["bucket_bytes"] = int(getattr(trainer.buckets, "bucket_bytes", 0)) or None
        except Exception as ex:                                    # the log format is RCCL's business: never fail the bench over it
            rccl_info = {"error": repr(ex)}
    if rank == 0:
        lv = trainer.loss_values()
        first = [k for k in lv if k not in ("l2", "total")][0]
        ce, ll, l2 = lv[first][0], lv[first][1], lv["l2"]
        # ---- roofline of the dominant kernel (HIP events recorded on the launch stream inside the timed region)
        # A bracket [e0, launch, e1] also times whatever the stream waited for between e0 and the launch: on a host-bound graph (DAN: ~1100
        # launches per step) the queue runs dry and a bracket can hold milliseconds of host time (profiles/r5/size1024_lines.jsonl line 1:
        # 2.5 ms "average" for a 30 us kernel).  Durations are therefore taken per LAYER as the MEDIAN over the region's steps (the k-th launch of a kernel
        # in every step is the same layer) and summed: a handful of inflated brackets no longer decide which kernel is "dominant" (VERDICT r5 6b).
        def robust_ms(evs, steps=None):
            # the k-th launch of a kernel in every step is the same layer: median over the steps, summed over the layers (a first version took the
            # median over all launches with equal FLOPs, which lumps conv2_2 / conv3_2 / conv4_2 — equal FLOPs, different durations — together
            # and read 10 % above rocprofv3's average)
            steps = steps or prof_steps
            per = len(evs) // steps if steps and len(evs) % steps == 0 else 0
            if not per:
                return sum(a.elapsed_time(b) for a, b, _ in evs)
            total = 0.0
            for j in range(per):
                v = sorted(evs[j + k * per][0].elapsed_time(evs[j + k * per][1]) for k in range(steps))
                total += v[len(v) // 2] * steps
            return total

        stats = []
        for label, evs in prof.items():
            ms = robust_ms(evs)
            fl = sum(f for _, _, f in evs)
            stats.append((ms, label, len(evs), fl))
        stats.sort(reverse=True)
        # The DOMINANT kernel is the one that carries the largest share of the step's FLOPs (round 6).  Ranking by bracketed time let a kernel
        # with a thousandth of the FLOPs take the slot whenever its brackets were inflated as a family (DAN at 1024 x 1024, eager: the 16-channel
        # flat-M head kernel "averaging" 37 ms inside a 40 ms step, frac 0.0000 - profiles/r6/size1024_lines.jsonl of the first collection);
        # for S3FD both rankings name conv_wgrad_rows_kernel<128>.  `kernels` below stays ranked by time, so such a family remains visible.
        ms, label, n, fl = max(stats, key=lambda t: t[3])
        achieved = fl / (ms * 1e-3) / 1e12
        # HBM bytes per launch from the committed PMC passes (tools/pmc_bench.sh; PMC collection needs its own rocprofv3 runs, so it cannot
        # happen inside this process).  The file is stamped with the hash of the kernel sources it was measured on: with other sources
        # the numbers describe another binary and `traffic` is null (the stale figure is reported beside it, flagged).
        traffic, traffic_stale, pmc = None, None, {}
        tj = next((c for c in (os.path.join(ROOT, "profiles", r, "pmc_bench_traffic.json") for r in ("r6", "r5", "r4", "r3")) if os.path.exists(c)),
                  os.path.join(ROOT, "profiles", "r6", "pmc_bench_traffic.json"))
        if os.path.exists(tj):
            pmc = json.load(open(tj))
            t = pmc.get(label)
            same = pmc.get("_meta", {}).get("csrc_sha256") == csrc_hash()
            if t and same:
                traffic = round(t["hbm_bytes_per_launch"])
            elif t:
                traffic_stale = {"bytes": round(t["hbm_bytes_per_launch"]), "why": os.path.relpath(tj, ROOT) + " was measured on other kernel sources"}
            if not same:
                pmc = {}
        calib = None        # library-GEMM peak measured on a box of this pool (tools/calibrate_peaks.py), next to the datasheet peak
        cj = os.path.join(ROOT, "profiles", "r1", "calibration.json")
        if os.path.exists(cj):
            cpk = max(json.load(open(cj)).get("gemm_bf16_4096_tflops", 0.0), json.load(open(cj)).get("gemm_bf16_8192_tflops", 0.0))
            if cpk > 0:
                calib = {"peak": cpk, "what": "library bf16 GEMM (torch.matmul) measured on this pool in round 1 (profiles/r1/calibration.json)", "frac": round(achieved / cpk, 4)}
        serial = None
        if prof_serial and label in prof_serial:
            sms = robust_ms(prof_serial[label], 2)
            sfl = sum(f for _, _, f in prof_serial[label])
            sach = sfl / (sms * 1e-3) / 1e12
            serial = {"achieved": round(sach, 2), "frac": round(sach / PEAK_BF16_TFLOPS, 4), "avg_launch_ms": round(sms / len(prof_serial[label]), 4),
                      "what": "same kernel with the weight-gradient stream off, 2 steps after the timed region"}
        roof = {"bound": "mfma", "kernel": label, "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_stale": traffic_stale,
                "traffic_over_algorithmic": (round(traffic / (sum(prof_bytes.get(label, [])) / max(1, len(prof_bytes.get(label, [])))), 3)
                                             if traffic and prof_bytes and prof_bytes.get(label) else None),
                "launches_per_step": n // prof_steps,
                "avg_launch_ms": round(ms / n, 4), "flop_per_launch": fl / n,
                "share_of_step_time": round(ms / prof_steps / (dt_prof / prof_steps * 1e3), 4), "calibrated": calib,
                "whole_step": {"what": "all convolution launches of the timed region (forward, data and weight gradients): algorithmic FLOP / wall time",
                               "tflop_per_step": round(sum(f for _, _, _, f in stats) / prof_steps / 1e12, 3),
                               "achieved": round(sum(f for _, _, _, f in stats) / prof_steps / (dt / args.steps) / 1e12, 1),
                               "frac": round(sum(f for _, _, _, f in stats) / prof_steps / (dt / args.steps) / 1e12 / PEAK_BF16_TFLOPS, 4)},
                "concurrency": ("weight-gradient kernels share the chip with the data-gradient kernels (second stream): in-region durations "
                                "include the shared time" if serial else None), "serialized": serial}
        # ---- HBM-bound convolutions of the timed region (1x1 lateral / context convs, Cout <= 16 heads, conv1_1): algorithmic bytes
        # (activations in + out, weights once) / event-bracketed duration against the 8 TB/s HBM3E peak
        hbm = []
        if prof_bytes:
            for m_, l_, n_, f_ in stats:
                nb = sum(prof_bytes.get(l_, []))
                if nb > 0 and f_ / nb < 310.0:                      # below the MFMA / HBM ridge (2.5 PFLOP/s / 8 TB/s)
                    hbm.append({"kernel": l_, "launches_per_step": n_ // prof_steps, "ms_per_step": round(m_ / prof_steps, 3),
                                "achieved": round(nb / (m_ * 1e-3) / 1e9, 1), "peak": 8000.0, "unit": "GB/s", "frac": round(nb / (m_ * 1e-3) / 8e12, 4),
                                "flop_per_byte": round(f_ / nb, 1), "algorithmic_bytes_per_launch": round(nb / n_),
                                "traffic": (round(pmc[l_]["hbm_bytes_per_launch"]) if l_ in pmc else None)})
        roof["hbm_bound_convs"] = hbm[:8]
        out = {
            "metric": "640x640 images/sec/node (train fwd+bwd)", "value": round(world * B * args.steps / dt, 3), "unit": "images/sec",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if args.global_batch else "weak", "vs_baseline": None, "dtype": _lib.ACT_NAME, "data": "synthetic",
            "config": {"workload": "%s, %dx%d %s training (fwd+bwd+SGD), batch %d per GPU" % (workload, S, S, _lib.ACT_NAME, B),
                       "global_batch": world * B, "parallelism": "dp%d" % world, "anchors_per_image": anchors.num_anchors,
                       "step_launch": "hipGraph replay" if args.graph else "eager",
                       "rccl_ranks": trainer.buckets.rccl.world if trainer.buckets.rccl is not None else 1,
                       "rccl": rccl_info,
                       "dp_transport": (("rccl %d called by libdanhip (danhip_comm_*), control plane %s" % (trainer.buckets.rccl.version, dist.get_backend() if multi else "none"))
                                        if trainer.buckets.rccl is not None else (trainer.buckets.transport if trainer.buckets.enabled else None)),
                       "dp_comm": (os.environ.get("DANHIP_DP_COMM", "allreduce") + "/" + os.environ.get("DANHIP_DP_BUCKET_DTYPE", "f32")) if trainer.buckets.enabled else None,
                       "weight_gradient_stream": bool((not trainer.buckets.enabled or trainer.buckets.device_collectives) and ops.WGRAD_STREAM)},
            "event_recording": {"steps": prof_steps, "ms_per_step": round(dt_prof / prof_steps * 1e3, 3), "launch": "eager",
                                "what": "the roofline region: the same steps repeated right after the timed region with two HIP events per convolution launch "
                                        "(the timed region records none); its own wall time per step shows what the events cost"},
            "repeats": {"regions": len(region_dts), "steps_each": args.steps, "ms_per_step": [round(d / args.steps * 1e3, 3) for d in region_dts],
                        "min_ms_per_step": round(min(region_dts) / args.steps * 1e3, 3),
                        "median_ms_per_step": round(sorted(region_dts)[len(region_dts) // 2] / args.steps * 1e3, 3),
                        "value_at_median": round(world * B * args.steps / sorted(region_dts)[len(region_dts) // 2], 3)},
            "strong": strong, "strong16": strong16,
            "loss": {"ce": round(ce, 4), "loc": round(ll, 4), "l2": round(l2, 4)},
            "roofline": roof,
            "kernels": [{"kernel": l, "ms_per_step": round(m / prof_steps, 3), "tflops": round(f / (m * 1e-3) / 1e12, 1)} for m, l, _, f in stats[:6]],
        }
        if eval_out:
            out["eval"] = eval_out
        if enc_out:
            out["target_encode"] = enc_out
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    shutdown_distributed(trainer)       # captured graph -> device drained -> RCCL communicator -> control-plane group


if __name__ == "__main__":
    main()
