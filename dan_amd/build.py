"""Builds libdanhip.so (bf16 activations) and libdanhip_f16.so (fp16 activations, -DDANHIP_FP16) — all HIP kernels + the
C ABI — in-tree for gfx950 with hipcc.

    python -m dan_amd.build [--force]

The .so is git-ignored but travels to the GPU box with the repo snapshot.
"""
import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libdanhip.so")
OUT_F16 = os.path.join(HERE, "libdanhip_f16.so")
OBJ = os.path.join(CSRC, "_obj")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-munsafe-fp-atomics", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function"]


def _newer(a, deps):
    return not os.path.exists(a) or any(os.path.getmtime(d) > os.path.getmtime(a) for d in deps)


def _build_variant(out, objdir, defines, force, verbose):
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.cpp")))
    hdrs = glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "danhip.h")]
    os.makedirs(objdir, exist_ok=True)
    jobs = []
    for s in srcs:
        o = os.path.join(objdir, os.path.basename(s) + ".o")
        if force or _newer(o, [s] + hdrs):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        extra = ["-ffp-contract=off"] if s.endswith("_exact.hip") else []   # bit-exact index/box kernels: no FMA contraction
        cmd = ["hipcc"] + FLAGS + defines + extra + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", s, "-o", o]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (s, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr)
        return o

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(cc, jobs))
    objs = [os.path.join(objdir, os.path.basename(s) + ".o") for s in srcs]
    if force or jobs or _newer(out, objs):
        r = subprocess.run(["hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs, capture_output=True, text=True)      # no vendor GEMM / conv library is linked: every kernel is in csrc/
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stderr)
    # An address-space cast the host pass rejects silently leaves a kernel's host stub undefined: catch it here.  The check runs in a
    # child process: loading the library HERE would bring the system HIP runtime into this process before PyTorch's own copy, and a
    # process that then uses torch.cuda (build() followed by smoke() in one interpreter) finds "no ROCm-capable device".
    r = subprocess.run([sys.executable, "-c", "import ctypes, sys; ctypes.CDLL(sys.argv[1])", out], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("%s does not load: %s" % (out, r.stderr.strip()[-2000:]))
    return out


def build(force=False, verbose=False):
    out = _build_variant(OUT, OBJ, [], force, verbose)
    _build_variant(OUT_F16, OBJ + "_f16", ["-DDANHIP_FP16"], force, verbose)
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
