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
    assert L.danhip_version() >= 2       # 2: danhip_deform_sample_bwd(..., workspace, workspace_bytes, stream)
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
    # ABI 2: danhip_deform_sample_bwd refuses a workspace sized to the version-1 contract (N*H*W*C floats, without the 64 statistic words)
    # before anything is launched (the pointers are never dereferenced on the host)
    L.danhip_deform_sample_bwd_workspace_bytes.restype = ctypes.c_size_t
    need = L.danhip_deform_sample_bwd_workspace_bytes(1, 8, 8, 64)
    assert need == (8 * 8 * 64 + 64) * 4
    buf = (ctypes.c_char * 64)()
    p = ctypes.cast(buf, ctypes.c_void_p)
    rc = _lib.lib().danhip_deform_sample_bwd(p, p, p, p, p, 1, 8, 8, 64, 3, 3, 1, 1, 1, 0, p, 8 * 8 * 64 * 4, None)
    assert rc == _lib.lib().danhip_deform_sample_bwd(p, p, p, p, p, 1, 8, 8, 64, 3, 3, 1, 1, 1, 0, p, need - 1, None) != 0
    assert b"workspace" in L.danhip_last_error()


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
# invariants of the overlap (trainer.GradBuckets.ready): one firing per variable and step, reverse creation order
b.begin_step()
b.ready("v5")
for bad, what in (("v5", "fired twice"), ):
    try:
        b.ready(bad); raise SystemExit("no error: " + what)
    except RuntimeError as e:
        assert what in str(e), e
b.begin_step()
b.ready("v2")
try:
    b.ready("v4"); raise SystemExit("no error: out of order")
except RuntimeError as e:
    assert "reverse creation order" in str(e), e
# DANHIP_DP_CHECK: a gradient written after its bucket was reduced is caught in finish()
os.environ["DANHIP_DP_CHECK"] = "1"
f.g = torch.full((f.total,), float(rank + 1))
b = GradBuckets(f, bucket_bytes=16 << 10)
b.begin_step()
for n in reversed(f.names):
    b.ready(n)
b.finish()                                  # clean step passes
f.g = torch.full((f.total,), float(rank + 1)); b.flat = f
b.begin_step()
b.ready("v5"); b.ready("v4")
f.g[f.starts[4] + 3] += 1.0                 # late contribution to v4
b.ready("v3"); b.ready("v2"); b.ready("v1"); b.ready("v0")
try:
    b.finish(); raise SystemExit("late write not detected")
except RuntimeError as e:
    assert "changed after its bucket" in str(e) and "'v4'" in str(e), e
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
'''


def _free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def test_gradient_buckets_allreduce_gloo_world2(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_fused_head_blocks_in_the_flat_buffer():
    """FlatParams lays loc_i | cls_i out as ONE block (VariableStore.fuse): the members are strided views of it (weights and
    gradients), the block is one optimizer segment, and TF-named export / load still see the separate variables."""
    import torch
    from dan_amd.net.variables import VariableStore
    from dan_amd.trainer import FlatParams, GradBuckets
    vs = VariableStore(device="cpu")
    a = vs.get("h/loc_0/kernel", (3, 3, 16, 4), "glorot")
    b = vs.get("h/loc_0/bias", (4,), "zeros")
    c = vs.get("h/cls_0/kernel", (3, 3, 16, 2), "glorot")
    d = vs.get("h/cls_0/bias", (2,), 0.5)
    vs.get("x/kernel", (1, 1, 8, 8), "glorot")
    a0, c0 = a.detach().clone(), c.detach().clone()
    kk, bb = ("h/loc_0/kernel", "h/cls_0/kernel"), ("h/loc_0/bias", "h/cls_0/bias")
    assert vs.fuse(kk, 3) is None and vs.fuse(bb, 0) is None           # not laid out yet: the caller concatenates
    flat = FlatParams(vs)
    W, B = vs.fuse(kk, 3), vs.fuse(bb, 0)
    assert W.shape == (3, 3, 16, 6) and W.is_contiguous() and torch.equal(B, torch.tensor([0, 0, 0, 0, .5, .5]))
    assert torch.equal(W[..., :4], a0) and torch.equal(W[..., 4:], c0) and torch.equal(a, a0) and torch.equal(c, c0)
    assert flat.names == ["h/loc_0/kernel", "h/loc_0/bias", "x/kernel"] and flat.sizes[:2] == [3 * 3 * 16 * 6, 6]
    W._danhip_grad[..., 5] = 7.0                                         # a gradient written into the block shows up in the member
    assert c.grad[..., 1].eq(7).all() and a.grad.eq(0).all()
    assert GradBuckets(flat).start_of["h/cls_0/kernel"] == flat.starts[0]
    assert set(n for n, _ in vs.named()) == {"h/loc_0/kernel", "h/loc_0/bias", "h/cls_0/kernel", "h/cls_0/bias", "x/kernel"}
    vs.load_tf_named({"h/cls_0/kernel": torch.ones(3, 3, 16, 2)})        # loading by TF name writes through the view
    assert W[..., 4:].eq(1).all() and torch.equal(W[..., :4], a0)
    assert torch.equal(vs.export_tf_named()["h/loc_0/bias"], torch.zeros(4))


def test_flat_params_block_diagonal_fuse():
    """VariableStore.fuse(..., "blockdiag") (DAN's stage-2 input mix, dan_amd/net/danet.py::_stage2_mix_fused): the two 1x1 kernels become the
    diagonal blocks of ONE [1, 1, 2C, C] segment of the flat buffers (zeros elsewhere), keep their TF names and shapes, their biases fuse
    along axis 0 right behind, and the gradient-bucket offsets still descend in backward order."""
    import torch
    from dan_amd.net.variables import VariableStore
    from dan_amd.trainer import FlatParams, GradBuckets
    C, c3 = 64, 21
    vs = VariableStore(device="cpu")
    vs.get("before/kernel", (1, 1, 8, 8), "glorot")
    w1 = vs.get("m/satge1_conv_1x1_0/kernel", (1, 1, C, c3), "glorot")
    vs.get("m/satge1_conv_1x1_0/bias", (c3,), 0.25)
    w2 = vs.get("m/residual_conv_1x1_0/kernel", (1, 1, C, C - c3), "glorot")
    vs.get("m/residual_conv_1x1_0/bias", (C - c3,), 0.5)
    vs.get("after/kernel", (1, 1, 8, 8), "glorot")
    w10, w20 = w1.detach().clone(), w2.detach().clone()
    kk = ("m/satge1_conv_1x1_0/kernel", "m/residual_conv_1x1_0/kernel")
    bb = ("m/satge1_conv_1x1_0/bias", "m/residual_conv_1x1_0/bias")
    assert vs.fuse(kk, "blockdiag") is None and vs.fuse(bb, 0) is None
    # gradient-free passes without a trainer (evaluation scripts): the block is built once, kept while its members are unchanged, and marked
    # for ops' packed-weight cache; VariableStore.build is the differentiable form plain autograd uses
    with torch.no_grad():
        t1 = vs.fuse(kk, "blockdiag")
        assert t1 is vs.fuse(kk, "blockdiag") and hasattr(t1, "_danhip_grad") and t1._danhip_grad is None and tuple(t1._danhip_lower.shape) == (1, 1, C, C)
        assert torch.equal(t1[0, 0, :C, :c3], w10[0, 0]) and torch.equal(t1[0, 0, C:, c3:], w20[0, 0]) and t1[0, 0, :C, c3:].eq(0).all() and t1[0, 0, C:, :c3].eq(0).all()
        w1.add_(1.0)
        t2 = vs.fuse(kk, "blockdiag")
        assert t2 is not t1 and torch.equal(t2[0, 0, :C, :c3], w10[0, 0] + 1.0)
        w1.copy_(w10)
    built = vs.build(kk, "blockdiag")
    assert built.requires_grad and torch.equal(built.detach()[0, 0, C:, c3:], w20[0, 0])
    built.sum().backward()
    assert w1.grad.eq(1).all() and w2.grad.eq(1).all()
    w1.grad = w2.grad = None
    flat = FlatParams(vs)
    assert "_infer_blocks" not in vs.__dict__
    wv, bv = vs.fuse(kk, "blockdiag"), vs.fuse(bb, 0)
    assert tuple(wv.shape) == (1, 1, 2 * C, C) and wv.is_contiguous() and tuple(bv.shape) == (C,)
    assert torch.equal(wv[0, 0, :C, :c3], w10[0, 0]) and torch.equal(wv[0, 0, C:, c3:], w20[0, 0])
    assert wv[0, 0, :C, c3:].eq(0).all() and wv[0, 0, C:, :c3].eq(0).all()
    assert torch.equal(w1, w10) and torch.equal(w2, w20) and tuple(w1.shape) == (1, 1, C, c3) and tuple(w2.shape) == (1, 1, C, C - c3)
    assert w1.data_ptr() == wv.data_ptr() and w2.data_ptr() == wv[0, 0, C:, c3:].data_ptr()
    low = wv._danhip_lower                                  # the second input's rows over ALL columns: what its data gradient multiplies by
    assert tuple(low.shape) == (1, 1, C, C) and low.is_contiguous() and low.data_ptr() == wv[0, 0, C:, :].data_ptr()
    assert flat.names == ["before/kernel", "m/satge1_conv_1x1_0/kernel", "m/satge1_conv_1x1_0/bias", "after/kernel"]
    assert flat.sizes[1:3] == [2 * C * C, C]
    wv._danhip_grad[0, 0, C + 3, c3 + 2] = 7.0              # a gradient written into the block shows up in the member
    assert w2.grad[0, 0, 3, 2].item() == 7.0 and w1.grad.eq(0).all()
    st = GradBuckets(flat).start_of
    assert st["m/residual_conv_1x1_0/kernel"] == st["m/satge1_conv_1x1_0/kernel"] == flat.starts[1]
    assert st["after/kernel"] > st["m/satge1_conv_1x1_0/bias"] > st["m/satge1_conv_1x1_0/kernel"] > st["before/kernel"]
    vs.load_tf_named({"m/residual_conv_1x1_0/kernel": torch.ones(1, 1, C, C - c3)})      # loading by TF name writes through the view
    assert wv[0, 0, C:, c3:].eq(1).all() and wv[0, 0, C:, :c3].eq(0).all() and torch.equal(wv[0, 0, :C, :c3], w10[0, 0])
    assert torch.equal(vs.export_tf_named()["m/satge1_conv_1x1_0/kernel"], w10)


def test_flat_params_deformable_kernel_as_its_gemm_operand():
    """VariableStore.fuse((name,), "hwio") (dan_amd/utility/custom_op.py::deform_conv_2d): the deformable convolution's OIHW variable is stored
    in the flat buffers as the [1, 1, kh * kw * C, Cout] GEMM operand its kernels consume; the TF variable is the permuted view (same name,
    shape and values; loading / exporting by TF name goes through the view; a gradient written into the block shows up in the variable)."""
    import torch
    from dan_amd.net.variables import VariableStore
    from dan_amd.trainer import FlatParams
    vs = VariableStore(device="cpu")
    vs.get("a/kernel", (1, 1, 8, 8), "glorot")
    k = vs.get("blk/deform_conv/kernel", (16, 8, 3, 3), "glorot_oihw")
    vs.get("blk/deform_conv/bias", (16,), "zeros")
    k0 = k.detach().clone()
    key = ("blk/deform_conv/kernel",)
    assert vs.fuse(key, "hwio") is None
    with torch.no_grad():                                # gradient-free pass without a trainer: built once, kept per weight version
        t = vs.fuse(key, "hwio")
        assert t is vs.fuse(key, "hwio") and tuple(t.shape) == (1, 1, 72, 16) and t._danhip_grad is None
        assert torch.equal(t.reshape(3, 3, 8, 16), k0.permute(2, 3, 1, 0))
    flat = FlatParams(vs)
    w1 = vs.fuse(key, "hwio")
    assert tuple(w1.shape) == (1, 1, 72, 16) and w1.is_contiguous() and torch.equal(w1.reshape(3, 3, 8, 16), k0.permute(2, 3, 1, 0))
    assert tuple(k.shape) == (16, 8, 3, 3) and torch.equal(k, k0) and not k.is_contiguous() and k.data_ptr() == w1.data_ptr()
    assert flat.names == ["a/kernel", "blk/deform_conv/kernel", "blk/deform_conv/bias"] and flat.sizes[1] == 16 * 8 * 9
    w1._danhip_grad[0, 0, 2 * 8 + 5, 7] = 3.0             # tap (0, 2), input channel 5, output channel 7
    assert k.grad[7, 5, 0, 2].item() == 3.0 and k.grad.abs().sum().item() == 3.0
    vs.load_tf_named({"blk/deform_conv/kernel": torch.full((16, 8, 3, 3), 2.0)})
    assert w1.eq(2).all()
    assert torch.equal(vs.export_tf_named()["blk/deform_conv/kernel"], torch.full((16, 8, 3, 3), 2.0))


def test_bench_spawns_its_own_ranks_when_typed_without_a_launcher():
    """`python bench.py --gpus 2` (no WORLD_SIZE): bench.py starts torch.distributed.run as a child before importing torch, the two ranks
    rendezvous on 127.0.0.1 and rank 0 prints the line (DANHIP_BENCH_DRY: the launch plumbing alone, gloo, no GPU)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["DANHIP_BENCH_DRY"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout
    import json
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["rccl_ranks"] == 2
    # the line carries BOTH series: `scaling` = "weak" for `value`, and a strong-scaling leg on the fixed global batch
    # of BASELINE.json configs[2] (128): 64 images per rank at N = 2, as four 16-image towers
    assert out["scaling"] == "weak" and out["strong_plan"] == [64, 4, 16]


def test_strong_scaling_plan_of_the_bench_line():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod_plan", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    assert b.strong_plan(128, 1) == (128, 8, 16) and b.strong_plan(128, 8) == (16, 1, 16) and b.strong_plan(128, 4) == (32, 2, 16)
    assert b.strong_plan(16, 8) == (2, 1, 2)             # a global batch of 16: 2-image towers
    assert b.strong_plan(128, 3) is None and b.strong_plan(0, 2) is None and b.strong_plan(48, 2) is None    # 24 per rank is not whole towers


def test_comm_entry_points_fail_with_a_message_when_rccl_or_the_communicator_is_missing():
    """include/danhip.h danhip_comm_*: librccl is bound at run time — a wrong path is DANHIP_ECOMM (-4) with dlopen's message, a call without
    a communicator is DANHIP_EINVAL; nothing throws, nothing needs a GPU."""
    code = r"""
import ctypes, sys
sys.path.insert(0, %r)
from dan_amd import _lib
L = _lib.lib()
assert L.danhip_comm_load(b"/nonexistent/librccl.so") == -4 and b"dlopen" in L.danhip_last_error()
assert L.danhip_comm_allreduce_sum(None, None, 4, 0, None) == -1 and b"communicator" in L.danhip_last_error()
assert L.danhip_comm_reduce_scatter_sum(None, None, None, 4, 0, None) == -1
assert L.danhip_comm_allgather(None, None, None, 4, 0, None) == -1
assert L.danhip_comm_destroy(None) == 0
assert L.danhip_comm_unique_id(None) == -1
print("OK")
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "OK" in r.stdout, (r.stdout, r.stderr[-2000:])


def _replay_pointwise_waits(NST, DPW, ksteps, n_items, stores_per_item, late):
    """Host replay of conv_pointwise_kernel's issue / wait order for ONE wave under in-order vmcnt retirement (csrc/conv_pointwise.hip:
    `step`): returns the list of (awaited DMA step, younger operations actually in the queue, count the kernel's rule allows to stay
    outstanding).  The rule is safe iff allowed <= younger at every wait."""
    floor = lambda n: 32 if n >= 32 else 24 if n >= 24 else 16 if n >= 16 else 12 if n >= 12 else 8 if n >= 8 else 6 if n >= 6 else 4 if n >= 4 else 2 if n >= 2 else 0
    queue = []                                     # issue order: ("dma", step) / ("st", item)
    d_idx = 0

    def issue():
        nonlocal d_idx
        queue.extend([("dma", d_idx)] * DPW)
        d_idx += 1
    for _ in range(NST - 1):
        issue()
    st_age, st_cnt, out, c_idx = NST, 0, [], 0
    for item in range(n_items):
        for k in range(ksteps):
            last = k == ksteps - 1
            pend = st_cnt if st_age <= NST - 2 - (1 if late else 0) else 0
            allowed = floor((NST - 2) * DPW + pend)
            pos = max(i for i, op in enumerate(queue) if op == ("dma", c_idx))
            out.append((c_idx, len(queue) - 1 - pos, allowed))
            if not late:
                issue()
            c_idx += 1
            st_age += 1
            if last:
                queue.extend([("st", item)] * stores_per_item)
                st_age, st_cnt = 0, stores_per_item
            if late:
                issue()
    return out


@pytest.mark.parametrize("bn,nst", [(256, 3), (128, 4), (64, 4)])
def test_pointwise_counted_vmcnt_never_lets_the_awaited_dma_stay_in_flight(bn, nst):
    """the waves that issue their DMA AFTER the epilogue stores (waves 4-7) must stop counting those stores as younger
    one K-step earlier than waves 0-3.  Replays both issue orders for every tile width and K depth the launcher uses."""
    dpw = 2 + bn // 64
    npt = {256: 4, 128: 2, 64: 1}[bn]
    for late in (False, True):
        for ksteps in (1, 2, 3, 4, 5, 9, 36):
            for op_idx, younger, allowed in _replay_pointwise_waits(nst, dpw, ksteps, 6, 2 * npt, late):
                assert allowed <= younger, (bn, nst, late, ksteps, op_idx, younger, allowed)


def test_four_gib_activations_are_refused_not_wrapped():
    """The convolution kernels build raw buffer descriptors with a 32-bit byte count: a 16 x 1024 x 1024 x 128 16-bit tensor is exactly 4 GiB
    and would wrap to a zero-length buffer.  Every entry point refuses it before any launch."""
    from dan_amd import _lib
    L = _lib.lib()
    d = _lib.ConvDesc()
    d.N, d.H, d.W, d.Cin, d.Cout, d.kh, d.kw, d.stride, d.Ho, d.Wo = 16, 1024, 1024, 128, 128, 3, 3, 1, 1024, 1024
    one = ctypes.c_void_p(16)                       # never dereferenced: the size check comes first
    assert L.danhip_conv2d_fwd(ctypes.byref(d), one, one, None, one, 1, 0, None, None) == -1 and b"2^31" in L.danhip_last_error()
    assert L.danhip_conv2d_bwd_data(ctypes.byref(d), one, one, None, one, 0, None) == -1 and b"2^31" in L.danhip_last_error()
    assert L.danhip_conv2d_bwd_weight(ctypes.byref(d), one, one, one, None, 128, None) == -1 and b"2^31" in L.danhip_last_error()
    d.N = 8                                          # the BASELINE per-GPU shard (batch 8 at 1024 x 1024) is 2 GiB: accepted by the check
    r, c = ctypes.c_int64(), ctypes.c_int64()
    assert L.danhip_conv_packed_dims(ctypes.byref(d), 0, ctypes.byref(r), ctypes.byref(c)) == 0


def test_bench_parses_an_rccl_info_log():
    """bench.py --gpus N describes its own collectives: rank 0 parses the RCCL INFO log of the run.  The parser is fed
    the line shapes RCCL / NCCL print (INIT: version / rank / nranks / channels; TUNING: algorithm and protocol per collective size; GRAPH:
    transports); missing lines leave fields empty rather than failing."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    log = """
box:123:123 [0] NCCL INFO RCCL version 2.22.3+hip7.0 HEAD:abcdef
box:123:140 [0] NCCL INFO comm 0x55 rank 0 nranks 8 cudaDev 0 nvmlDev 0 busId c000 commId 0x1 - Init START
box:123:140 [0] NCCL INFO Channel 00/16 :    0   1   2   3   4   5   6   7
box:123:140 [0] NCCL INFO Channel 00 : 0[0] -> 1[1] via P2P/IPC
box:123:140 [0] NCCL INFO 16 coll channels, 16 collnet channels, 0 nvls channels, 16 p2p channels, 2 p2p channels per peer
box:123:140 [0] NCCL INFO AllReduce: 33554432 Bytes -> Algo 1 proto 2 time 412.5
box:123:140 [0] NCCL INFO AllReduce: 4096 Bytes -> Algo 0 proto 0 time 9.1
"""
    info = b.rccl_debug_parse(log)
    assert info["version"].startswith("2.22") and info["nranks"] == 8 and info["channels"] == 16
    assert info["algo"]["AllReduce/32MiB"] == "Ring" and info["proto"]["AllReduce/32MiB"] == "Simple"
    assert info["algo"]["AllReduce/4096B"] == "Tree" and info["proto"]["AllReduce/4096B"] == "LL"
    assert info["transport"] == ["P2P/IPC"]
    # the line shapes RCCL 2.26 (ROCm 7.0 wheel) prints on the GPU boxes, one-rank group
    info = b.rccl_debug_parse("""
runc:1:1 [0] NCCL INFO RCCL version : 2.26.6-HEAD:64f48b6
runc:1:2 [0] NCCL INFO Ring 126 : 0 -> 0 -> 0 comm 0x63 nRanks 01 busId a7000
runc:1:2 [0] NCCL INFO Ring 127 : 0 -> 0 -> 0 comm 0x63 nRanks 01 busId a7000
runc:1:2 [0] NCCL INFO 128 coll channels, 128 collnet channels, 0 nvls channels, 64 p2p channels, 128 p2p channels per peer
runc:1:2 [0] NCCL INFO ncclCommInitRankConfig_impl comm 0x63 rank 0 nranks 1 cudaDev 0 nvmlDev 0 busId a7000 commId 0x52 - Init COMPLETE
""")
    assert info["version"].startswith("2.26.6") and info["nranks"] == 1 and info["channels"] == 128 and info["rings"] == 2
    empty = b.rccl_debug_parse("nothing here")
    assert empty["nranks"] is None and empty["algo"] == {}


def test_fused_context_block_fires_its_gradient_hooks_in_flat_buffer_order():
    """The data-parallel buckets need every variable's 'gradient ready' hook in non-increasing order of its offset in the flat buffer
    (GradBuckets.ready).  The fused DAN context block (net/danet.py:842-918 as one autograd node) issues all of its gradients and then fires
    the hooks itself; three of its 1x1 kernels live side by side in ONE segment (VariableStore.fuse) that stands where the first of them was
    created.  Replays variable creation, the flat layout and the block's hook order on the CPU - no kernel runs."""
    from dan_amd.net.variables import VariableStore
    from dan_amd.trainer import FlatParams
    vs = VariableStore(device="cpu", seed=1)
    c = 256
    order = [("branch1_conv_1x1", 1, 1, c, 64), ("branch2_conv_1x1", 1, 1, c, 64), ("branch3_conv_1x1", 1, 1, c, 64), ("branch3_conv_3x1", 3, 1, 64, 32),
             ("branch3_conv_1x3", 1, 3, 64, 32), ("branch4_conv_1x1", 1, 1, c, 64), ("branch4_conv_3x3", 3, 3, 64, 64), ("branch4_conv_1x3", 3, 1, 64, 32),
             ("branch4_conv_3x1", 1, 3, 64, 32), ("residual_conv", 1, 1, 256, c)]
    vs.get("before/kernel", (3, 3, 8, 64), "glorot")
    for blk in ("s1", "s2"):
        for scope, kh, kw, ci, co in order:
            vs.get("%s/%s/kernel" % (blk, scope), (kh, kw, ci, co), "glorot")
            vs.get("%s/%s/bias" % (blk, scope), (co,), "zeros")
        for suffix, axis in (("kernel", 3), ("bias", 0)):
            assert vs.fuse(tuple("%s/%s/%s" % (blk, s, suffix) for s in ("branch3_conv_1x1", "branch4_conv_1x1", "branch2_conv_1x1")), axis=axis) is None
        for va, vb in (("branch3_conv_3x1", "branch3_conv_1x3"), ("branch4_conv_1x3", "branch4_conv_3x1")):      # the 3x1 | 1x3 pairs as "plus" kernels
            assert vs.fuse(("%s/%s/kernel" % (blk, va), "%s/%s/kernel" % (blk, vb)), axis="plus") is None
            assert vs.fuse(("%s/%s/bias" % (blk, va), "%s/%s/bias" % (blk, vb)), axis=0) is None
    vs.get("after/kernel", (3, 3, 64, 64), "glorot")
    plus_built = vs.build(("s1/branch3_conv_3x1/kernel", "s1/branch3_conv_1x3/kernel"), "plus").detach().clone()      # the torch-op form (plain autograd)
    flat = FlatParams(vs)
    assert torch.equal(plus_built, vs.fused[("s1/branch3_conv_3x1/kernel", "s1/branch3_conv_1x3/kernel")])
    start = flat.start_of_member
    # the block's hook order (dan_amd/net/danet.py::_se_inception_block_fused), blocks in reverse creation order as backward visits them
    hook_scopes = ["residual_conv", "branch4_conv_3x1", "branch4_conv_1x3", "branch4_conv_3x3", "branch3_conv_1x3", "branch3_conv_3x1", "branch4_conv_1x1",
                   "branch3_conv_1x1", "branch2_conv_1x1", "branch1_conv_1x1"]
    seq = [start["after/kernel"]]
    for blk in ("s2", "s1"):
        seq += [start["%s/%s/kernel" % (blk, s)] for s in hook_scopes]
    seq.append(start["before/kernel"])
    assert all(a >= b for a, b in zip(seq, seq[1:])), seq
    # firing a kernel's hook declares its bias final too: every bias sits at or above its kernel's segment
    for blk in ("s1", "s2"):
        for scope, *_ in order:
            assert start["%s/%s/bias" % (blk, scope)] >= start["%s/%s/kernel" % (blk, scope)]
    # the three fused 1x1 kernels are views of one block laid out [b3 | b4 | b2] along the output-channel axis
    wcat = vs.fused[("s1/branch3_conv_1x1/kernel", "s1/branch4_conv_1x1/kernel", "s1/branch2_conv_1x1/kernel")]
    assert tuple(wcat.shape) == (1, 1, c, 192) and wcat.is_contiguous()
    w4 = dict(vs.named())["s1/branch4_conv_1x1/kernel"]
    assert w4.data_ptr() == wcat[..., 64:128].data_ptr() and not w4.is_contiguous()
    # each 3x1 | 1x3 pair: one [3, 3, 64, 64] block, the 3x1 member in the middle column (outputs 0..31), the 1x3 member in the middle row
    named = dict(vs.named())
    for va, vb in (("s1/branch3_conv_3x1", "s1/branch3_conv_1x3"), ("s2/branch4_conv_1x3", "s2/branch4_conv_3x1")):
        blkw = vs.fused[(va + "/kernel", vb + "/kernel")]
        a3, b3 = named[va + "/kernel"], named[vb + "/kernel"]
        assert tuple(blkw.shape) == (3, 3, 64, 64) and blkw.is_contiguous() and tuple(a3.shape) == (3, 1, 64, 32) and tuple(b3.shape) == (1, 3, 64, 32)
        assert a3.data_ptr() == blkw[:, 1:2, :, 0:32].data_ptr() and b3.data_ptr() == blkw[1:2, :, :, 32:64].data_ptr()
        assert torch.equal(blkw[:, 1:2, :, 0:32], a3) and torch.equal(blkw[1:2, :, :, 32:64], b3)
        rest = blkw.clone()
        rest[:, 1:2, :, 0:32] = 0
        rest[1:2, :, :, 32:64] = 0
        assert rest.eq(0).all()                                   # zeros everywhere else
        assert tuple(vs.fused[(va + "/bias", vb + "/bias")].shape) == (64,)
    # the gradient of a whole block also fills the zeros' places: mask_structured clears exactly those
    assert len(flat.struct_grads) == 4
    flat.g.fill_(1.0)
    flat.mask_structured()
    g3 = vs.fused[("s1/branch3_conv_3x1/kernel", "s1/branch3_conv_1x3/kernel")]._danhip_grad
    assert g3.sum().item() == 2 * 3 * 64 * 32 and named["s1/branch3_conv_3x1/kernel"].grad.eq(1).all() and named["s1/branch3_conv_1x3/kernel"].grad.eq(1).all()
    assert named["s1/branch4_conv_3x3/kernel"].grad.eq(1).all() and wcat._danhip_grad.eq(1).all()      # nothing else is touched
    # cleared by selection, not by a 0 / 1 multiply: an overflowed gradient (fp16 build, static loss scale) at a constant-zero place must not
    # become 0 * inf = NaN in momentum and weight
    g3[0, 0, 5, 7] = float("inf")
    g3[2, 2, 1, 40] = float("nan")
    flat.mask_structured()
    assert torch.isfinite(g3).all() and g3.sum().item() == 2 * 3 * 64 * 32


def test_gradient_buckets_split_a_small_tail_off_the_last_bucket():
    """trainer.GradBuckets (round 5): the last bucket holds the FIRST layers' variables, whose gradients backward produces at the very end of
    the step - it is split so that only a small tail (<= tail_bytes) is reduced / updated behind the last backward kernel.  S3FD-like
    layout: the buckets still tile the flat buffer back to front, on segment boundaries."""
    from dan_amd.trainer import GradBuckets

    class Flat:
        pass
    f = Flat()
    sizes = [1728, 64, 36864, 64, 73728, 128, 147456, 128, 294912, 256, 589824, 256, 589824, 256, 1179648, 512, 2359296, 512, 2359296, 512,
             2359296, 512, 2359296, 512, 2359296, 512, 4718592, 1024, 1048576, 1024]
    f.starts, off = [], 0
    for n in sizes:
        f.starts.append(off); off += (n + 63) // 64 * 64
    f.total = off
    f.names = ["v%d" % i for i in range(len(sizes))]
    f.g = torch.zeros(8)                                   # (CPU: no streams; only the layout is exercised)
    b = GradBuckets(f, bucket_bytes=32 << 20, tail_bytes=2 << 20)
    cov = sorted(b.bounds)
    assert cov[0][0] == 0 and cov[-1][1] == f.total and all(cov[i][1] == cov[i + 1][0] for i in range(len(cov) - 1))
    assert all(s in f.starts for s, _ in b.bounds)
    assert b.bounds == sorted(b.bounds, reverse=True)      # launch order = back to front
    s_tail, e_tail = b.bounds[-1]
    assert s_tail == 0 and 0 < (e_tail - s_tail) * 4 <= 2 << 20
    assert (b.bounds[-2][1] - b.bounds[-2][0]) * 4 > 2 << 20
    # a buffer smaller than two tails keeps its single bucket
    g = Flat()
    g.starts, g.total, g.names, g.g = [0, 64000], 128000, ["a", "b"], torch.zeros(8)
    assert GradBuckets(g, bucket_bytes=32 << 20, tail_bytes=2 << 20).bounds == [(0, 128000)]


FALLBACK_WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from dan_amd import trainer
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
# rank 1 cannot load RCCL (simulated); rank 0 could: BOTH must come back with None, neither may wait in a collective the other skipped
real = trainer.call
def fake(name, *a):
    if name == "danhip_comm_load" and rank == 1:
        raise RuntimeError("librccl.so: cannot open shared object file (simulated)")
    if name == "danhip_comm_create":
        raise AssertionError("no rank may try to build the communicator after one of them failed to load the library")
    return real(name, *a)
trainer.call = fake
got = trainer.RcclComm.shared_or_none(torch.device("cpu"))
assert got is None, got
dist.barrier(); dist.destroy_process_group()
print("ok", rank)
'''


def test_every_rank_falls_back_to_the_process_group_when_one_cannot_have_rccl(tmp_path):
    """A multi-GPU job whose RCCL communicator cannot be built on some rank keeps running on the control plane's collectives (and says so
    on stderr) instead of hanging or dying: the decision is taken by all ranks together (dan_amd.trainer.RcclComm.shared_or_none)."""
    script = tmp_path / "f.py"
    script.write_text(FALLBACK_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    assert all("gradients travel on the process group" in o for o in outs), outs


def test_one_rank_without_rccl_is_an_error_not_a_silent_fallback(monkeypatch):
    import torch
    from dan_amd import trainer
    monkeypatch.setattr(trainer.RcclComm, "_shared", None)
    assert trainer.RcclComm.shared_or_none(torch.device("cpu")) is None            # no GPU here: the communicator cannot exist
    monkeypatch.setenv("DANHIP_DP_NO_FALLBACK", "1")
    with pytest.raises(Exception):
        trainer.RcclComm.shared_or_none(torch.device("cpu"))


def test_ops_steering_state_lives_in_a_context_object(monkeypatch):
    """dan_amd.ops keeps its kernel-form switches, diagnostic sinks and per-step hooks in ONE OpsContext; the rounds-1-4 spelling
    (`ops.USE_SLOTS = False`) reads and writes the ACTIVE context, `ops.use_context` scopes another one."""
    from dan_amd import ops
    base = ops.context()
    assert ops.USE_SLOTS is base.USE_SLOTS is True and ops.TRACE is None
    assert "USE_SLOTS" not in vars(ops) and "TRACE" not in vars(ops) and "GRAD_READY_HOOK" not in vars(ops)     # no module-level copies
    mine = ops.OpsContext(USE_SPLITK=False)
    with ops.use_context(mine):
        assert ops.context() is mine and ops.USE_SPLITK is False and base.USE_SPLITK is True
        ops.TRACE = {"x": 1}                                   # legacy spelling writes the active context only
        assert mine.TRACE == {"x": 1} and base.TRACE is None
        assert ops._conv_scratch.__module__ == "dan_amd.ops"
    assert ops.context() is base and ops.TRACE is None and ops.USE_SPLITK is True
    monkeypatch.setattr(ops, "USE_SLOTS", False)               # pytest's monkeypatch goes through the same shim (and restores through it)
    assert base.USE_SLOTS is False and ops._new_slot(True) is None
    monkeypatch.undo()
    assert base.USE_SLOTS is True
    with pytest.raises(AttributeError):
        ops.OpsContext(NO_SUCH_SWITCH=1)
    with pytest.raises(AttributeError):
        ops.NO_SUCH_ATTRIBUTE
    ops.SOME_NEW_MODULE_ATTRIBUTE = 3                           # anything that is not a context field stays an ordinary module attribute
    assert vars(ops)["SOME_NEW_MODULE_ATTRIBUTE"] == 3
    del ops.SOME_NEW_MODULE_ATTRIBUTE


# ---- the RCCL bootstrap cannot hang a job: tests/ddp/fake_rccl.cpp stands in for librccl; its
# ncclCommInitRank needs no GPU, and with FAKE_RCCL_INIT_HANG=1 it waits for its peers for ever, as the real one does.
BOOTSTRAP_WORKER = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch, torch.distributed as dist
from dan_amd import trainer
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
scenario = os.environ["SCENARIO"]
dist.init_process_group("gloo", rank=rank, world_size=world)
if scenario == "dies_before_bootstrap" and rank == 1:
    os._exit(17)                                     # gone before anybody reaches the communicator
if scenario == "never_calls_create" and rank == 1:
    real = trainer.call
    def stuck(name, *a):
        if name == "danhip_comm_create":
            time.sleep(1000)                         # alive, but never enters ncclCommInitRank
        return real(name, *a)
    trainer.call = stuck
comm = trainer.RcclComm.shared_or_none(torch.device("cpu"))
assert scenario == "healthy", "a rank got past a bootstrap that cannot complete"
import ctypes
n, r, d = ctypes.c_int32(), ctypes.c_int32(), ctypes.c_int32()
trainer.call("danhip_comm_info", comm.handle, ctypes.byref(n), ctypes.byref(r), ctypes.byref(d))
assert (n.value, r.value) == (world, rank) and comm.version == 29999 and comm.async_error() == 0
comm.close()
trainer.shutdown_distributed()
print("bootstrap ok", rank)
'''


def _bootstrap(tmp_path, scenario, hang):
    import time
    from tests_ddp_paths import FAKE_RCCL
    script = tmp_path / "b.py"
    script.write_text(BOOTSTRAP_WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), WORLD_SIZE="2", SCENARIO=scenario, DANHIP_RCCL_PATH=FAKE_RCCL,
               DANHIP_COMM_TIMEOUT_S="6", FAKE_RCCL_INIT_HANG="1" if hang else "0", DANHIP_DP_NO_FALLBACK="1")
    t0 = time.time()
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    return procs, outs, time.time() - t0


def test_two_ranks_bootstrap_a_communicator_through_the_stand_in_library(tmp_path):
    """unique id on rank 0 -> (ok, id) broadcast over the control plane -> ncclCommInitRank(2) under its deadline -> info / async error /
    destroy -> orderly shutdown, on CPU (no collective: those need a device)."""
    procs, outs, _ = _bootstrap(tmp_path, "healthy", hang=False)
    assert all(p.returncode == 0 for p in procs) and all("bootstrap ok" in o for o in outs), outs


def test_a_rank_that_never_enters_comm_create_ends_the_job_within_the_deadline(tmp_path):
    """One rank is alive but never calls danhip_comm_create; the other waits inside ncclCommInitRank, which (like the real library) has
    no timeout of its own: after DANHIP_COMM_TIMEOUT_S every rank ends with trainer.DEADLINE_EXIT_CODE and says why."""
    procs, outs, took = _bootstrap(tmp_path, "never_calls_create", hang=True)
    assert [p.returncode for p in procs] == [75, 75], outs
    assert all("FATAL" in o and "ncclCommInitRank" in o and "did not return within" in o for o in outs), outs
    assert took < 60, took


def test_a_rank_that_died_before_the_bootstrap_is_noticed_at_the_control_barrier(tmp_path):
    procs, outs, took = _bootstrap(tmp_path, "dies_before_bootstrap", hang=True)
    assert procs[1].returncode == 17 and procs[0].returncode == 75, outs
    assert "FATAL" in outs[0] and "control-plane barrier" in outs[0], outs
    assert took < 60, took
