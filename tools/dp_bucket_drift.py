#!/usr/bin/env python3
"""Is sending the gradient buckets as bf16 (DANHIP_DP_BUCKET_DTYPE=bf16: half the xGMI bytes, PyramidBox moves 819 MB per step) harmless?

VERDICT r2 item 9: before bf16 buckets become a recommendation, show over >= 200 steps that the 2^-9-per-element rounding of each
reduced gradient does not move the trajectory more than the run-to-run noise the fp32 path already has (the weight-gradient kernels
combine their split partial sums in a data-dependent order).  One GPU is enough for that question: a forced ONE-rank RCCL group runs the
real bucket code path (cast -> collective -> cast back), and with one rank the collective adds nothing, so the ONLY difference between the
two arms is the rounding.

    python tools/dp_bucket_drift.py --model pb --steps 200 [--size 320 --batch 4]

Runs three child processes from the same seed on a fixed set of synthetic batches (a different batch every step, cycled): fp32 buckets
twice (A, A': the noise floor) and bf16 buckets once (B); prints one JSON line with, every 25 steps, the loss of each arm and the relative
parameter distances |wB - wA| / |wA - w0| and |wA' - wA| / |wA - w0| (distance travelled from the initial point as the yardstick)."""
import argparse
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(args):
    import torch
    from dan_amd import synthetic
    from dan_amd.trainer import init_distributed
    init_distributed()
    dev = torch.device("cuda", 0)
    S, B = args.size, args.batch
    nb = 8                                                     # distinct batches, cycled
    if args.model == "pb":
        from dan_amd.train_pb import PBAnchorTargets, PBModel, PBTrainer
        model = PBModel(device=dev)
        tr = PBTrainer(model, world=1, base_lr=args.lr, lr_boundaries=(10 ** 9,), lr_factors=(1.0, 1.0))
        enc = PBAnchorTargets(S, S, dev)
        batches = [(synthetic.make_images(B, S, S, dev, seed=100 + i), (enc.encode_batch(synthetic.make_gt_boxes(B, S, S, seed=200 + i)),)) for i in range(nb)]
    else:
        from dan_amd.train_sfd import AnchorConfig, SFDModel, SFDTrainer
        model = SFDModel(device=dev)
        tr = SFDTrainer(model, world=1, base_lr=args.lr, lr_boundaries=(10 ** 9,), lr_factors=(1.0, 1.0))
        anchors = AnchorConfig(S, S, dev)
        batches = []
        for i in range(nb):
            loc_t, cls_t, _ = anchors.encode_batch(synthetic.make_gt_boxes(B, S, S, seed=200 + i))
            batches.append((synthetic.make_images(B, S, S, dev, seed=100 + i), (loc_t, cls_t)))
    assert tr.buckets.enabled and tr.buckets.device_collectives, "the forced one-rank RCCL group is not active"
    w0 = tr.flat.w.clone()
    rec = {"w0": w0.cpu(), "w": {}, "loss": {}}
    for step in range(1, args.steps + 1):
        img, tg = batches[(step - 1) % nb]
        tr.train_step(img, *tg)
        if step % args.every == 0 or step == args.steps:
            rec["w"][step] = tr.flat.w.detach().cpu().clone()
            rec["loss"][step] = tr.loss_values()["total"]
    torch.save(rec, args.out)
    import torch.distributed as dist
    dist.barrier()
    dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="pb", choices=["pb", "sfd"])
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--every", type=int, default=25)
    ap.add_argument("--size", type=int, default=320)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--out", default="")
    ap.add_argument("--worker", action="store_true")
    args = ap.parse_args()
    if args.worker:
        return worker(args)
    import torch
    tmp = tempfile.mkdtemp()
    outs = {}
    def free_port():                      # (as bench.py's spawn_ranks: never a fixed port another job on the box may hold)
        import socket
        with socket.socket() as so:
            so.bind(("127.0.0.1", 0))
            return so.getsockname()[1]

    for arm, wire in (("fp32_a", "f32"), ("fp32_b", "f32"), ("bf16", "bf16")):
        port = free_port()
        env = dict(os.environ, DANHIP_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                   DANHIP_DP_BUCKET_DTYPE=wire)
        env.pop("DANHIP_DIST_BACKEND", None)
        out = os.path.join(tmp, arm + ".pt")
        cmd = [sys.executable, os.path.abspath(__file__), "--worker", "--out", out, "--model", args.model, "--steps", str(args.steps),
               "--every", str(args.every), "--size", str(args.size), "--batch", str(args.batch), "--lr", str(args.lr)]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True)
        if r.returncode != 0:
            raise SystemExit("%s failed:\n%s\n%s" % (arm, r.stdout[-2000:], r.stderr[-4000:]))
        outs[arm] = torch.load(out)
    a, a2, b = outs["fp32_a"], outs["fp32_b"], outs["bf16"]
    assert torch.equal(a["w0"], b["w0"]) and torch.equal(a["w0"], a2["w0"])
    rows = []
    for step in sorted(a["w"]):
        moved = (a["w"][step] - a["w0"]).norm().item()
        rows.append({"step": step, "loss_fp32": round(a["loss"][step], 4), "loss_fp32_rerun": round(a2["loss"][step], 4), "loss_bf16": round(b["loss"][step], 4),
                     "moved_from_init": moved, "bf16_vs_fp32_rel": (b["w"][step] - a["w"][step]).norm().item() / moved,
                     "fp32_rerun_vs_fp32_rel": (a2["w"][step] - a["w"][step]).norm().item() / moved})
    print(json.dumps({"what": "one-rank RCCL group, %s %dx%d batch %d, lr %g, %d steps: bf16 gradient buckets against fp32 buckets and against the fp32 "
                              "path's own run-to-run noise; distances relative to the distance travelled from the initial parameters"
                              % (args.model, args.size, args.size, args.batch, args.lr, args.steps), "rows": rows}))


if __name__ == "__main__":
    main()
