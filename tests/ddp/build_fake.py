"""Builds tests/ddp/libfake_rccl.so from fake_rccl.cpp (TEST INFRASTRUCTURE: the nccl* entry points over shared memory, so that the
product's RCCL transport runs with several ranks on one GPU).  Called by __graft_entry__.build() so the file travels to the GPU box."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "fake_rccl.cpp")
OUT = os.path.join(HERE, "libfake_rccl.so")


def build(force=False):
    if force or not os.path.exists(OUT) or os.path.getmtime(OUT) < os.path.getmtime(SRC):
        r = subprocess.run(["hipcc", "-O2", "-fPIC", "-shared", "-std=c++17", "-x", "hip", "--offload-arch=gfx950", SRC, "-o", OUT, "-lrt"],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (SRC, r.stderr))
    return OUT


if __name__ == "__main__":
    print(build(force=True))
