#!/usr/bin/env python3
"""Finds compiler-inserted LDS-DMA queue drains in the hand-pipelined kernels: an `s_waitcnt vmcnt(N)` that hipcc (not an asm statement of ours)
placed right in front of LDS reads or a workgroup barrier.  The kernels keep their LDS-DMA prefetch in flight across barriers with COUNTED
asm waits; a compiler wait there - hipcc orders a ds_read behind every builtin LDS-DMA it has seen, and protects the destination registers
of conditionally consumed loads at loop headers - drains the whole queue (round 5: conv_halo.hip had one per nine steps in both wave groups
and one per item in the bit-mask data gradient).

    python tools/audit_compiler_drains.py [file.hip ...]      (default: the pipelined convolution kernels)
Prints every hit with its kernel; exit code 1 if any kernel named in EXPECT_CLEAN has one."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "dan_amd", "csrc")
DEFAULT = ["conv_halo.hip", "conv_wgrad_rows.hip", "conv_wgrad_pw.hip", "conv_pointwise.hip", "conv_halo_c64.hip", "deform_fused.hip"]
EXPECT_CLEAN = ["conv3x3_halo_kernelILi8ELi32ELi128ELi4ELi2ELi1ELi4E", "conv3x3_halo_kernelILi16ELi16ELi128ELi4ELi2ELi1ELi4E"]


def scan(path):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-fno-gpu-rdc", "-S", "--cuda-device-only", "-o", out, path],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-2000:])
        txt = open(out).read().split("\n")
    hits, name, inasm = [], None, False
    for i, l in enumerate(txt):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            name = m.group(1)
        if "ASMSTART" in l:
            inasm = True
        if "ASMEND" in l:
            inasm = False
        if name and not inasm and re.search(r"s_waitcnt.*vmcnt\(\d+\)", l):
            ctx = [t.strip() for t in txt[i + 1:i + 8]]
            loop = any("in Loop" in t for t in txt[max(0, i - 40):i])
            if loop and any(("ds_read" in t or "s_barrier" in t) for t in ctx):
                hits.append((name, l.strip(), [t for t in ctx if t][:2]))
    return hits


def main():
    files = sys.argv[1:] or DEFAULT
    bad = 0
    for f in files:
        path = f if os.path.isabs(f) else os.path.join(CSRC, f)
        hits = scan(path)
        print("%s: %d compiler wait(s) in front of LDS reads / barriers inside loops" % (os.path.basename(path), len(hits)))
        for name, w, ctx in hits:
            print("   %-90s %s | %s" % (name[:90], w, " ; ".join(ctx)))
            if any(k in name for k in EXPECT_CLEAN):
                bad += 1
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
