#!/usr/bin/env python3
"""Same-box, alternating A/B of bench.py between library builds and / or environment switches (cdna_hip_programming.md rule 24: never rank
builds by timings from different devices).  Every arm is `name=ENV1=v1,ENV2=v2` (empty = the default library); the arms run in turn for
R rounds, each run a fresh bench.py process; prints per arm the median / min ms per step over all its timed regions.

    python tools/ab_bench.py [--rounds 3] [--steps 30] [--bench-args "..."] base= prev=DANHIP_LIB_PATH=dan_amd/libdanhip_prev.so
"""
import argparse
import json
import os
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--bench-args", default="")
    ap.add_argument("arms", nargs="+")
    a = ap.parse_args()
    arms = []
    for spec in a.arms:
        name, _, envs = spec.partition("=")
        env = dict(kv.split("=", 1) for kv in envs.split(",") if kv)
        arms.append((name, env))
    ms = {n: [] for n, _ in arms}
    for r in range(a.rounds):
        for name, env in arms:
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", str(a.steps), "--no-cpu-baseline", "--no-eval", "--strong-global-batch", "0", "--strong16-global-batch", "0",
                   "--no-serialized-roofline", "--events-steps", "1"] + a.bench_args.split()
            out = subprocess.run(cmd, env=dict(os.environ, **env), capture_output=True, text=True)
            lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
            if out.returncode != 0 or not lines:
                print("arm %s failed:\n%s" % (name, out.stderr[-2000:]))
                sys.exit(1)
            d = json.loads(lines[-1])
            ms[name].extend(d["repeats"]["ms_per_step"])
            print("round %d %-10s %s" % (r, name, d["repeats"]["ms_per_step"]), flush=True)
    base = statistics.median(ms[arms[0][0]])
    for name, _ in arms:
        med = statistics.median(ms[name])
        print("%-10s median %.3f ms  min %.3f ms  (%+.2f %% vs %s)" % (name, med, min(ms[name]), 100.0 * (med / base - 1.0), arms[0][0]))


if __name__ == "__main__":
    main()
