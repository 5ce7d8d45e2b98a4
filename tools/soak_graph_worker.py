#!/usr/bin/env python3
"""Soak of the data-parallel step captured as ONE hipGraph (tests/ddp/worker.py, DDP_MODE=graph, forced one-rank RCCL communicator):
N consecutive fresh child processes, each must exit 0 and land on the same parameters (fp32-atomics noise).  Prints one line per run
and a summary; exit code 1 if any run failed.  VERDICT r4 item 1a: 30/30 on the GPU box, log kept under profiles/.

    python tools/soak_graph_worker.py [N=30] [model=sfd]
"""
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = os.path.join(ROOT, "tests", "ddp", "worker.py")


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    model = sys.argv[2] if len(sys.argv) > 2 else "sfd"
    import torch
    env = dict(os.environ, DANHIP_FORCE_DIST="1", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", DDP_MODE="graph", DDP_MODEL=model)
    env.pop("DANHIP_DP_TRANSPORT", None)
    ref, bad = None, 0
    with tempfile.TemporaryDirectory() as tmp:
        for i in range(n):
            out = os.path.join(tmp, "g%d.pt" % i)
            t0 = time.time()
            r = subprocess.run([sys.executable, WORKER, out], env=env, capture_output=True, text=True, timeout=600)
            dt = time.time() - t0
            if r.returncode != 0:
                bad += 1
                print("run %2d FAILED rc=%d in %.1fs\n--- stderr head\n%s\n--- stderr tail\n%s" % (i, r.returncode, dt, r.stderr[:4000], r.stderr[-4000:]), flush=True)
                continue
            w = torch.load(out)["w"]
            if ref is None:
                ref = w
            dev = (w - ref).abs().max().item() / ref.abs().max().item()
            ok = dev <= 1e-4
            bad += 0 if ok else 1
            print("run %2d ok rc=0 %.1fs  max|w - w_run0| / max|w| = %.2e%s" % (i, dt, dev, "" if ok else "  DIVERGED"), flush=True)
    print("SOAK %s: %d / %d runs green (model %s, graph-captured step incl. bucketed RCCL all-reduce, one-rank communicator)" % ("OK" if bad == 0 else "FAILED", n - bad, n, model))
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
