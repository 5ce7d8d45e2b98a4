"""CPU suite: the C-ABI library loads and exports every symbol include/danhip.h declares (no compute without a GPU);
host-side logic (schedules, flat parameter layout, data-parallel bucket all-reduce over gloo, world size 2)."""
import ctypes
import os
import re
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from dan_amd import _lib, build
    build.build()
    L = _lib.lib()
    hdr = open(os.path.join(ROOT, "include", "danhip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = sorted(set(re.findall(r"\b(danhip_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 25, names
    for n in names:
        assert hasattr(L, n), "missing export " + n
    for n in _lib.SIGNATURES:
        assert n in names, "ctypes table binds %s which the header does not declare" % n
    assert L.danhip_version() >= 1
    assert L.danhip_act_dtype() == 1
    # the fp16 build of the same sources exports the same ABI
    L16 = ctypes.CDLL(build.OUT_F16)
    for n in names:
        assert hasattr(L16, n), "fp16 build misses export " + n
    L16.danhip_act_dtype.restype = ctypes.c_int
    assert L16.danhip_act_dtype() == 2


def test_invalid_arguments_are_reported_not_thrown():
    from dan_amd import _lib
    L = _lib.lib()
    d = _lib.ConvDesc()
    d.N, d.H, d.W, d.Cin, d.Cout, d.kh, d.kw, d.stride, d.Ho, d.Wo = 1, 8, 8, 3, 8, 3, 3, 1, 8, 8     # Cin not a multiple of 8
    r, c = ctypes.c_int64(), ctypes.c_int64()
    rc = L.danhip_conv_packed_dims(ctypes.byref(d), 0, ctypes.byref(r), ctypes.byref(c))
    assert rc == -1 and b"multiple of 8" in L.danhip_last_error()
    d.Cin = 64
    assert L.danhip_conv_packed_dims(ctypes.byref(d), 0, ctypes.byref(r), ctypes.byref(c)) == 0
    assert (r.value, c.value) == (16, 576)
    assert L.danhip_conv_kernel_label(ctypes.byref(d), 0) == b"conv_igemm_kernel<64, 16, 1, true>"
    # small_mining_match attribute validation mirrors the op constructor (small_mining_match.cc:291-306)
    rc = L.danhip_small_mining_match(None, 1, 1, 0.0, 0.4, 0.4, 6, 0.3, None, None, None, 0, None)
    assert rc == -1


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from dan_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "SO_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.DanhipError):
        _lib.lib()


def test_lr_schedule():
    from dan_amd.trainer import lr_schedule
    assert lr_schedule(0) == pytest.approx(1e-4) and lr_schedule(1000) == pytest.approx(1e-4)
    assert lr_schedule(1001) == pytest.approx(1e-3) and lr_schedule(80001) == pytest.approx(1e-4)
    assert lr_schedule(100001) == pytest.approx(1e-5)


WORKER = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from dan_amd.trainer import GradBuckets
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
class Flat: pass
f = Flat()
sizes = [1000, 64, 5000, 128, 30000, 64]
f.starts, off = [], 0
for n in sizes:
    f.starts.append(off); off += (n + 63) // 64 * 64
f.total = off
f.names = ["v%%d" %% i for i in range(len(sizes))]
f.g = torch.full((f.total,), float(rank + 1))
b = GradBuckets(f, bucket_bytes=16 << 10)
assert b.enabled and len(b.bounds) >= 2
covered = sorted(b.bounds)
assert covered[0][0] == 0 and covered[-1][1] == f.total and all(covered[i][1] == covered[i + 1][0] for i in range(len(covered) - 1))
b.begin_step()
for n in reversed(f.names):      # backward order
    b.ready(n)
b.finish()
assert torch.all(f.g == float(sum(range(1, world + 1)))), f.g[:4]
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
'''


def test_gradient_buckets_allreduce_gloo_world2(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
