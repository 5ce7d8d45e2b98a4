"""Training harness shared by the S3FD / PyramidBox / DAN model_fns: flat fp32 parameter / gradient / momentum
buffers, fused momentum-SGD (+L2 term, bias-gradient x2) and data-parallel gradient all-reduce over RCCL.

Replaces (semantics only) tf_replicate_model_fn.py:297-343,458-498,615-645 of the reference: each rank computes the
gradient of loss_rank / N on its contiguous shard of the global batch; gradients are summed across ranks (one
all-reduce per bucket of the flat gradient buffer, launched on a side stream as soon as the bucket's last gradient
has been produced so it overlaps the remaining backward kernels); the update is applied identically on every rank.
"""
import atexit
import ctypes
import os
import sys
import weakref

import torch
import torch.distributed as dist

from ._lib import call, ptr, stream


def comm_timeout_s():
    """Deadline (seconds) of every point where a rank waits for its peers outside a stream: the control-plane barriers around the RCCL
    bootstrap, ncclCommInitRank itself, and the watchdog's wait for a step's collectives (GradBuckets.finish).  DANHIP_COMM_TIMEOUT_S."""
    return float(os.environ.get("DANHIP_COMM_TIMEOUT_S", "300"))


DEADLINE_EXIT_CODE = 75          # (EX_TEMPFAIL) a rank that gave up waiting for its peers leaves with this status


def _die(msg):
    """A peer is gone or stuck: this rank cannot leave the RCCL call (or the stream) it waits in, so the PROCESS ends — message on stderr,
    non-zero status, no destructors (they would wait for the device), never a re-exec.  The launcher (torch.distributed.run) then takes the
    other ranks down; a supervisor restarts the job from its last checkpoint as a fresh set of processes."""
    sys.stderr.write("dan_amd: FATAL (rank %s): %s\n" % (os.environ.get("RANK", "0"), msg))
    sys.stderr.flush()
    os._exit(DEADLINE_EXIT_CODE)


def _with_deadline(fn, seconds, what):
    """fn() on a helper thread (ctypes releases the GIL inside the library); the calling thread waits `seconds` for it and ends the
    process otherwise (a thread blocked inside ncclCommInitRank cannot be cancelled)."""
    import threading
    box = {}

    def run():
        try:
            box["v"] = fn()
        except BaseException as e:                  # noqa: BLE001 - re-raised on the calling thread
            box["e"] = e

    th = threading.Thread(target=run, name="danhip-comm-bootstrap", daemon=True)
    th.start()
    th.join(seconds)
    if th.is_alive():
        _die("%s did not return within %.0f s (DANHIP_COMM_TIMEOUT_S): a peer rank never entered it or died inside it" % (what, seconds))
    if "e" in box:
        raise box["e"]
    return box.get("v")


_control = {"group": None, "made": False}


def control_group():
    """The process group that carries CONTROL traffic (CPU tensors, pickled objects, monitored barriers): the default group when its
    backend is gloo, else a gloo group over the same ranks created once (collective: every rank reaches this at the same point of
    GradBuckets / RcclComm construction).  A default group on the nccl backend (DANHIP_DIST_BACKEND=nccl, or a caller's own
    init_process_group) cannot carry CPU tensors: ADVICE r5."""
    if not (dist.is_available() and dist.is_initialized()):
        return None
    if not _control["made"]:
        _control["made"] = True
        if dist.get_backend() != "gloo":
            import datetime
            _control["group"] = dist.new_group(backend="gloo", timeout=datetime.timedelta(seconds=max(comm_timeout_s(), 1.0)))
    return _control["group"]


def control_barrier(what):
    """All ranks are alive and HERE, or this rank ends within the deadline: gloo's monitored barrier names the missing rank."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return
    import datetime
    try:
        dist.monitored_barrier(group=control_group(), timeout=datetime.timedelta(seconds=comm_timeout_s()), wait_all_ranks=False)
    except Exception as e:                          # noqa: BLE001 - any failure of the control plane here means a peer is gone
        _die("control-plane barrier %s failed: %s" % (what, str(e).splitlines()[0] if str(e) else type(e).__name__))


def rccl_library_path():
    """DANHIP_RCCL_PATH (a test's stand-in, tests/ddp/fake_rccl.cpp, or another build) or the copy PyTorch ships — that one is (or will
    be) in this process: bind that file rather than a second copy from the loader path.  None: the loader path."""
    env = os.environ.get("DANHIP_RCCL_PATH")
    if env:
        return env.encode()
    cand = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return cand.encode() if os.path.exists(cand) else None


class RcclComm(object):
    """This rank's RCCL communicator behind include/danhip.h's danhip_comm_* (csrc/comm.cpp): collectives are plain asynchronous calls on
    the CURRENT torch stream — no process-group work objects, no watchdog thread, capturable into a hipGraph like a kernel launch.

    torch.distributed (gloo, see init_distributed / control_group) is the CONTROL plane only: it carries the 128-byte unique id from rank 0
    to the others and the barriers around the bootstrap.  One communicator per process is shared by every trainer (RcclComm.shared).

    No step of the bootstrap can hang a job: a monitored control-plane barrier stands in front of it
    (a rank that died earlier is noticed there), rank 0 broadcasts (ok, id) so a failing ncclGetUniqueId is seen by every rank,
    ncclCommInitRank runs under a deadline on a helper thread, and a second barrier stands behind it."""
    _shared = None
    _live = weakref.WeakSet()

    def __init__(self, rank, world, device_index):
        call("danhip_comm_load", rccl_library_path())
        ident = ctypes.create_string_buffer(128)
        if world > 1:
            control_barrier("before the RCCL bootstrap")
            box = [None]
            if rank == 0:
                try:
                    call("danhip_comm_unique_id", ident)
                    box = [(True, ident.raw)]
                except Exception as e:              # noqa: BLE001 - every rank must learn of it: they all wait in the broadcast below
                    box = [(False, "%s: %s" % (type(e).__name__, e))]
            dist.broadcast_object_list(box, src=0, group=control_group())
            ok, payload = box[0]
            if not ok:
                raise RuntimeError("rank 0 could not produce the RCCL unique id: %s" % payload)
            ident = ctypes.create_string_buffer(payload, 128)
        else:
            call("danhip_comm_unique_id", ident)
        h = ctypes.c_void_p()

        def create():
            call("danhip_comm_create", ident, world, rank, int(device_index), ctypes.byref(h))

        if world > 1:
            _with_deadline(create, comm_timeout_s(), "ncclCommInitRank (danhip_comm_create, %d ranks)" % world)
        else:
            create()
        self.handle, self.rank, self.world = h, rank, world
        v = ctypes.c_int(0)
        call("danhip_comm_rccl_version", ctypes.byref(v))
        self.version = v.value
        RcclComm._live.add(self)

    @classmethod
    def shared(cls, device):
        if cls._shared is None or cls._shared.handle is None:
            rank = dist.get_rank() if dist.is_initialized() else 0
            world = dist.get_world_size() if dist.is_initialized() else 1
            index = -1 if device.type != "cuda" else (device.index if device.index is not None else torch.cuda.current_device())
            cls._shared = cls(rank, world, index)
        return cls._shared

    @classmethod
    def shared_or_none(cls, device):
        """The shared communicator, or None on EVERY rank when it could not be built on any of them (librccl missing from the process,
        communicator initialisation refused): the caller then keeps the gradients on the process group's host-staged collectives — slow,
        but a multi-GPU job still runs and says so (`transport`).  The ranks agree through the control plane after each local step that
        can fail, so none of them waits in a collective the others never enter.  DANHIP_DP_NO_FALLBACK=1: raise instead."""
        if cls._shared is not None and cls._shared.handle is not None:
            return cls._shared
        multi = dist.is_initialized() and dist.get_world_size() > 1
        strict = os.environ.get("DANHIP_DP_NO_FALLBACK", "0") == "1"

        def everyone(ok, what):
            if not multi:
                return ok
            control_barrier(what)                    # (a rank that died since the last agreement ends the job here, within the deadline)
            flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=control_group())
            return bool(flag.item())

        err = None
        try:                                         # (1) the library is loadable here (a local step: no rank waits on another)
            call("danhip_comm_load", rccl_library_path())
        except Exception as e:                       # noqa: BLE001 - reported below
            err = e
        if not everyone(err is None, "after loading librccl"):
            if strict and err is not None:
                raise err
            sys.stderr.write("dan_amd: RCCL is not loadable on every rank (%s): gradients travel on the process group (gloo)\n" % (err,))
            return None
        try:                                         # (2) the communicator itself (collective inside RCCL, under a deadline)
            comm = cls.shared(device)
        except Exception as e:                       # noqa: BLE001
            err, comm = e, None
        if not everyone(comm is not None, "after ncclCommInitRank"):
            if comm is not None:
                comm.close()
            if strict and err is not None:
                raise err
            sys.stderr.write("dan_amd: no RCCL communicator on every rank (%s): gradients travel on the process group (gloo)\n" % (err,))
            return None
        return comm

    @staticmethod
    def _dtype(t):
        return {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}[t.dtype]

    def all_reduce(self, t):
        call("danhip_comm_allreduce_sum", self.handle, ptr(t), t.numel(), self._dtype(t), stream())

    def reduce_scatter(self, shard, full):
        call("danhip_comm_reduce_scatter_sum", self.handle, ptr(full), ptr(shard), shard.numel(), self._dtype(full), stream())

    def all_gather(self, full, shard):
        call("danhip_comm_allgather", self.handle, ptr(shard), ptr(full), shard.numel(), self._dtype(full), stream())

    def async_error(self):
        """0 while healthy, else the RCCL error code of a collective that failed asynchronously (ncclCommGetAsyncError)."""
        if self.handle is None:
            return 0
        e = ctypes.c_int32(0)
        call("danhip_comm_async_error", self.handle, ctypes.byref(e))
        return e.value

    def abort(self):
        """Tear down WITHOUT waiting for outstanding collectives (ncclCommAbort): the only way out when a peer is gone."""
        if self.handle is not None:
            h, self.handle = self.handle, None
            try:
                call("danhip_comm_abort", h)
            except Exception:                        # noqa: BLE001 - the process is on its way out
                pass
        if RcclComm._shared is self:
            RcclComm._shared = None

    def close(self):
        """Every stream that carries this communicator's collectives must have drained, and every hipGraph holding them must be gone."""
        if self.handle is not None:
            h, self.handle = self.handle, None
            if torch.cuda.is_available():
                torch.cuda.synchronize()
            call("danhip_comm_destroy", h)
        if RcclComm._shared is self:
            RcclComm._shared = None


class CommWatch(object):
    """What ProcessGroupNCCL's watchdog thread did, done by the training loop itself (with the collectives issued straight on
    RCCL nothing notices a dead peer — the host runs ahead, the device waits in ncclAllReduce for ever).

    tick(stream) is called once per step behind the step's last collective: it records an event there and polls
    ncclCommGetAsyncError (one C call).  The host may run at most `lag` steps ahead of the device: before it records step k's event it
    requires step k - lag's event to have completed, polling that event and the async error until DANHIP_COMM_TIMEOUT_S — in the normal
    case the event is long done and the check costs one hipEventQuery.  On an error or on expiry the communicator is aborted and the
    process ends non-zero (a fresh set of processes resumes from the checkpoint; never a re-exec)."""

    def __init__(self, comm, lag=2):
        self.comm, self.lag, self.events = comm, lag, []

    def _check_error(self):
        e = self.comm.async_error() if self.comm is not None else 0
        if e != 0:
            from ._lib import lib
            msg = lib().danhip_last_error().decode(errors="replace")
            self.comm.abort()
            _die("asynchronous RCCL error %d in the gradient exchange: %s" % (e, msg))

    def wait(self, ev, what):
        import time
        deadline = time.monotonic() + comm_timeout_s()
        pause = 1e-4
        while not ev.query():
            self._check_error()
            if time.monotonic() > deadline:
                if self.comm is not None:
                    self.comm.abort()
                _die("%s did not complete within %.0f s (DANHIP_COMM_TIMEOUT_S): a peer rank is gone or stuck" % (what, comm_timeout_s()))
            time.sleep(pause)
            pause = min(pause * 2, 0.05)
        self._check_error()

    def tick(self, stream_):
        self._check_error()
        if len(self.events) >= self.lag:
            self.wait(self.events.pop(0), "the gradient exchange of an earlier step")
        ev = torch.cuda.Event()
        ev.record(stream_)
        self.events.append(ev)

    def drain(self):
        """Before shutdown: every recorded step has finished (or the process ends)."""
        while self.events:
            self.wait(self.events.pop(0), "the gradient exchange of the last steps")


@atexit.register
def _close_comms():          # a communicator must not outlive the HIP runtime: close what the program left open, before interpreter teardown
    for c in list(RcclComm._live):
        try:
            c.close()
        except Exception:
            pass


def lr_schedule(step, base_lr=1e-3, boundaries=(1000, 80000, 100000), factors=(0.1, 1.0, 0.1, 0.01), end_lr=1e-6):
    """tf.train.piecewise_constant + floor — train_sfd.py:429-434 (flags :98-109)."""
    i = 0
    while i < len(boundaries) and step > boundaries[i]:
        i += 1
    return max(base_lr * factors[i], end_lr)


class FlatParams(object):
    """Packs every variable of a VariableStore into one contiguous fp32 buffer (same for gradients and momenta) and
    re-points the nn.Parameters (and their .grad / gradient sinks) at views of it."""

    def __init__(self, vs, weight_decay=5e-4):
        named = vs.named()
        dev = named[0][1].device
        byname = dict(named)
        vs.__dict__.pop("_infer_blocks", None)       # blocks cached for gradient-free passes (VariableStore.fuse): the members move below
        # fused blocks (VariableStore.fuse): the members become strided views of ONE segment, placed where the first member stood
        group_of, members_done = {}, set()
        for key, axis in getattr(vs, "fuse_groups", []):
            if all(k in byname for k in key):
                group_of[key[0]] = (key, axis)
                members_done.update(key[1:])
        units = []                                   # (segment name, [member names], axis or None, numel)
        for name, p in named:
            if name in members_done:
                continue
            if name in group_of:
                key, axis = group_of[name]
                if axis == "blockdiag":              # two HWIO kernels of DIFFERENT inputs as one block-diagonal kernel over [x1 | x2]
                    p1, p2 = byname[key[0]], byname[key[1]]
                    numel = p1.shape[0] * p1.shape[1] * (p1.shape[2] + p2.shape[2]) * (p1.shape[3] + p2.shape[3])
                elif axis == "plus":                 # a k x 1 and a 1 x k kernel of the SAME input as one k x k kernel (columns side by side)
                    p1, p2 = byname[key[0]], byname[key[1]]
                    numel = p1.shape[0] * p2.shape[1] * p1.shape[2] * (p1.shape[3] + p2.shape[3])
                else:
                    numel = sum(byname[k].numel() for k in key)
                units.append((name, list(key), axis, numel))
            else:
                units.append((name, [name], None, p.numel()))
        sizes = [u[3] for u in units]
        # every segment starts on a 64-element (256-byte) boundary so views stay 16-byte aligned
        starts, off = [], 0
        for n in sizes:
            starts.append(off)
            off += (n + 63) // 64 * 64
        self.total = off
        self.w = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.g = torch.zeros(self.total, dtype=torch.float32, device=dev)
        self.v = torch.zeros(self.total, dtype=torch.float32, device=dev)
        gm, wd = [], []
        self.struct_grads, self.struct_masks = [], []     # "plus" blocks: gradient views and their 0 / 1 patterns (mask_structured)
        self.struct_zero_idx = []                         # ... and the flat-buffer offsets of their constant-zero places
        self.names, self.starts, self.sizes = [], starts, sizes
        self.start_of_member = {}
        for (name, members, axis, n), s in zip(units, starts):
            if axis is None:
                p = byname[name]
                self.w[s:s + n].copy_(p.data.reshape(-1))
                p.data = self.w[s:s + n].view(p.shape)
                p.grad = self.g[s:s + n].view(p.shape)
                p._danhip_grad = p.grad
            elif axis == "blockdiag":
                # [kh, kw, c1 + c2, o1 + o2] with the members on the diagonal and zeros elsewhere: the zeros get no gradient (ops._ConcatMix
                # clears the off-diagonal blocks of the weight gradient), so weight decay and momentum keep them at exactly zero
                p1, p2 = [byname[k] for k in members]
                kh, kw, c1, o1 = p1.shape
                shape = [kh, kw, c1 + p2.shape[2], o1 + p2.shape[3]]
                wblock, gblock = self.w[s:s + n].view(shape), self.g[s:s + n].view(shape)
                for q, (r0, r1, q0, q1) in ((p1, (0, c1, 0, o1)), (p2, (c1, shape[2], o1, shape[3]))):
                    wblock[:, :, r0:r1, q0:q1].copy_(q.data)
                    q.data = wblock[:, :, r0:r1, q0:q1]
                    q.grad = gblock[:, :, r0:r1, q0:q1]
                    q._danhip_grad = q.grad
                wblock._danhip_grad = gblock
                wblock._danhip_members = [p1, p2]
                wblock._danhip_blockdiag = (c1, o1)
                lower = wblock[:, :, c1:, :]                 # [kh, kw, c2, o1 + o2]: the second input's kernel over ALL output columns (its
                lower._danhip_grad = gblock[:, :, c1:, :]    # data gradient reads the full output gradient: no ragged column slice)
                lower._danhip_members = [p2]
                wblock._danhip_lower = lower
                vs.fused[tuple(members)] = wblock
            elif axis == "hwio":
                # ONE member stored transposed: the deformable convolution's OIHW variable (utility/custom_op.py:134) lives in the flat
                # buffers as the HWIO GEMM operand [kh * kw * C, Cout] its kernels consume (cached packing, weight gradient written in
                # place); the TF variable is the permuted VIEW of it - no transposing copy per step in either direction
                (p1,) = [byname[k] for k in members]
                co, ci, kh, kw = p1.shape
                wblock, gblock = self.w[s:s + n].view(kh, kw, ci, co), self.g[s:s + n].view(kh, kw, ci, co)
                wblock.copy_(p1.data.permute(2, 3, 1, 0))
                p1.data = wblock.permute(3, 2, 0, 1)
                p1.grad = gblock.permute(3, 2, 0, 1)
                p1._danhip_grad = p1.grad
                w1 = wblock.view(1, 1, kh * kw * ci, co)
                w1._danhip_grad = gblock.view(1, 1, kh * kw * ci, co)
                w1._danhip_members = [p1]
                vs.fused[tuple(members)] = w1
            elif axis == "plus":
                # [k, k, c, o1 + o2]: the k x 1 member in the middle column for outputs 0 .. o1-1, the 1 x k member in the middle row for the
                # other outputs, zeros elsewhere.  A weight gradient computed for the whole block also fills the zeros' places:
                # mask_structured() clears them before the optimizer (weight decay and momentum then keep the zeros exactly zero)
                p1, p2 = [byname[k] for k in members]
                k, one, c, o1 = p1.shape
                assert one == 1 and tuple(p2.shape[:3]) == (1, k, c) and k % 2 == 1, (p1.shape, p2.shape)
                mid = k // 2
                shape = [k, k, c, o1 + p2.shape[3]]
                wblock, gblock = self.w[s:s + n].view(shape), self.g[s:s + n].view(shape)
                mask = torch.zeros(shape, dtype=torch.float32, device=dev)
                for q, sl in ((p1, (slice(None), slice(mid, mid + 1), slice(None), slice(0, o1))),
                              (p2, (slice(mid, mid + 1), slice(None), slice(None), slice(o1, shape[3])))):
                    wblock[sl].copy_(q.data)
                    q.data = wblock[sl]
                    q.grad = gblock[sl]
                    q._danhip_grad = q.grad
                    mask[sl] = 1.0
                wblock._danhip_grad = gblock
                wblock._danhip_members = [p1, p2]
                self.struct_grads.append(gblock)
                self.struct_masks.append(mask)
                self.struct_zero_idx.append((mask.reshape(-1) == 0).nonzero().reshape(-1) + s)
                vs.fused[tuple(members)] = wblock
            else:
                ps = [byname[k] for k in members]
                ax = axis % ps[0].dim()
                shape = list(ps[0].shape)
                shape[ax] = sum(q.shape[ax] for q in ps)
                wblock, gblock = self.w[s:s + n].view(shape), self.g[s:s + n].view(shape)
                lo = 0
                for q in ps:
                    hi = lo + q.shape[ax]
                    wblock.narrow(ax, lo, hi - lo).copy_(q.data)
                    q.data = wblock.narrow(ax, lo, hi - lo)
                    q.grad = gblock.narrow(ax, lo, hi - lo)
                    q._danhip_grad = q.grad
                    lo = hi
                wblock._danhip_grad = gblock
                wblock._danhip_members = ps
                vs.fused[tuple(members)] = wblock
            for k in members:
                self.start_of_member[k] = s
            # gradient multipliers (train_sfd.py:436-439) and the L2 term (train_sfd.py:419-427)
            gm.append(2.0 if "/bias" in name else 1.0)
            if "bn" in name:                           # train_sfd.py:421 ('bn' not in trainable_var.name)
                wd.append(0.0)
            elif "l2_norm_layer" in name:
                wd.append(0.2 * weight_decay)
            elif "/bias" in name:
                wd.append(0.0)
            else:
                wd.append(weight_decay)
            self.names.append(name)
        self.seg = torch.tensor(starts + [self.total], dtype=torch.int64, device=dev)
        self.gmult = torch.tensor(gm, dtype=torch.float32, device=dev)
        self.wdc = torch.tensor(wd, dtype=torch.float32, device=dev)
        self.l2 = torch.zeros(1, dtype=torch.float32, device=dev)
        self.struct_zero_idx = torch.cat(self.struct_zero_idx) if self.struct_zero_idx else None

    def zero_grad(self):
        self.g.zero_()

    def mask_structured(self):
        """Clears the gradient entries that stand where a structured block ("plus") holds constant zeros - ONE launch for all blocks, by
        SELECTION (index_fill_ over the precomputed flat offsets), not by multiplying with a 0 / 1 pattern: 0 * inf is NaN, and in the fp16
        build with a static loss scale an overflowed weight gradient at such a place would reach momentum and weight of an entry that must
        stay exactly 0.  Linear, so it commutes with the data-parallel sum: once per step, after the all-reduce, before the
        optimizer."""
        if self.struct_zero_idx is not None:
            self.g.index_fill_(0, self.struct_zero_idx, 0.0)

    def range_plan(self, s, e):
        """(first segment, segment count, device table of the segment starts relative to s, flat offsets of the structured zeros inside) of
        the flat range [s, e) - which must begin and end on segment boundaries (gradient buckets do)."""
        k0, k1 = self.starts.index(s), (self.starts.index(e) if e < self.total else len(self.starts))
        rel = (self.seg[k0:k1 + 1] - s).contiguous()
        idx = None
        if self.struct_zero_idx is not None:
            m = (self.struct_zero_idx >= s) & (self.struct_zero_idx < e)
            idx = self.struct_zero_idx[m].contiguous() if bool(m.any()) else None
        return k0, k1 - k0, rel, idx

    def sgd_range(self, s, e, plan, lr, momentum=0.9, grad_scale=1.0):
        """The update of sgd_step restricted to the flat range [s, e) (plan = range_plan(s, e)), on the CURRENT stream: the same kernel on
        offset pointers and a relative segment table.  The caller zeroes self.l2 once per step and refreshes the packings of the range."""
        k0, nseg, rel, idx = plan
        if idx is not None:
            self.g.index_fill_(0, idx, 0.0)
        off = ctypes.c_void_p
        base = lambda t, elems: off(t.data_ptr() + 4 * elems)
        call("danhip_sgd_momentum_flat", base(self.w, s), base(self.g, s), base(self.v, s), ptr(rel), base(self.gmult, k0), base(self.wdc, k0), nseg,
             e - s, float(lr), float(momentum), float(grad_scale), ptr(self.l2), stream())

    def sgd_step(self, lr, momentum=0.9, grad_scale=1.0, dynamic_state=None):
        """dynamic_state: fp32[4] device tensor {loss scale, clean steps, growth interval, flag} (danhip_sgd_momentum_flat_dynamic)."""
        self.mask_structured()
        self.l2.zero_()
        if dynamic_state is not None:
            call("danhip_sgd_momentum_flat_dynamic", ptr(self.w), ptr(self.g), ptr(self.v), ptr(self.seg), ptr(self.gmult), ptr(self.wdc),
                 len(self.names), self.total, float(lr), float(momentum), ptr(dynamic_state), ptr(self.l2), stream())
        else:
            call("danhip_sgd_momentum_flat", ptr(self.w), ptr(self.g), ptr(self.v), ptr(self.seg), ptr(self.gmult), ptr(self.wdc), len(self.names),
                 self.total, float(lr), float(momentum), float(grad_scale), ptr(self.l2), stream())
        from . import ops
        ops.WEIGHT_EPOCH += 1          # the kernel wrote the parameters through raw pointers: cached bf16 packings are stale
        ops.repack_all()               # ... and are refreshed by one launch for all conv weights


class GradBuckets(object):
    """Bucketed all-reduce of the flat gradient buffer, overlapped with backward.

    Variables are laid out in forward (creation) order, so backward completes them from the END of the buffer towards
    the start; bucket b is reduced once backward has produced every gradient at or after its start offset.  The
    trainer calls `ready(name)` from per-layer autograd hooks; reductions run on a dedicated side stream."""

    def __init__(self, flat, bucket_bytes=32 << 20, tail_bytes=2 << 20):
        self.flat = flat
        # on_bucket(b, s, e): called on the communication stream behind bucket b's reduction (the trainer's per-bucket optimizer);
        # local = True runs the bucket machinery (events, side stream, on_bucket) without any collective: one GPU, overlapped optimizer
        self.on_bucket = None
        self.local = False
        self.enabled = (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1) or (os.environ.get("DANHIP_FORCE_DIST") == "1" and flat.g.is_cuda)
        # DANHIP_FAKE_ALLREDUCE=1 (diagnosis, single process): run the bucket machinery with a device-only stand-in for the collective
        self.fake = (not self.enabled) and os.environ.get("DANHIP_FAKE_ALLREDUCE") == "1" and flat.g.is_cuda
        self.enabled = self.enabled or self.fake
        self.bucket_bytes = bucket_bytes
        per = max(1, bucket_bytes // 4)
        self.bounds = []
        end = flat.total
        # walk segments backwards, closing a bucket once it holds >= per elements
        cur_end = end
        for s in reversed(flat.starts):
            if cur_end - s >= per:
                self.bounds.append((s, cur_end))
                cur_end = s
        if cur_end > 0:
            self.bounds.append((0, cur_end))
        # The LAST bucket holds the first layers' variables, whose gradients backward produces at the very end of the step: whatever is
        # reduced (and updated) behind them is exposed.  It is therefore split so that its final part is small (<= tail_bytes: conv1_1 ..
        # conv3_1 of the VGG backbone, 2.2 MB) - the larger part above it becomes complete ~3 ms of backward earlier and hides there.
        s0, e0 = self.bounds[-1]
        tail = max(1, tail_bytes // 4)
        if e0 - s0 > 2 * tail:
            cut = max((st for st in flat.starts if s0 < st <= s0 + tail), default=None)
            if cut is not None and cut < e0:
                self.bounds[-1:] = [(cut, e0), (s0, cut)]
        self.on_gpu = flat.g.is_cuda
        # Data plane: RCCL called directly through the library (RcclComm) for GPU buffers — DANHIP_DP_TRANSPORT=gloo keeps the process
        # group's host-staged collectives instead (the test hook that runs several ranks on ONE GPU: RCCL refuses duplicate devices).
        # Device-side collectives coexist with the weight-gradient stream; gloo's host-staged ones stalled with it.
        self.transport = dp_transport() if (self.enabled and not self.fake and self.on_gpu) else ("fake" if self.fake else "gloo")
        self.rccl = RcclComm.shared_or_none(flat.g.device) if self.transport == "rccl" else None
        if self.transport == "rccl" and self.rccl is None:
            self.transport = "gloo"                  # (agreed by all ranks inside shared_or_none; a one-rank forced job has no group: stays local)
            if not (dist.is_initialized() and dist.get_world_size() > 1):
                raise RuntimeError("RCCL communicator unavailable and no process group to fall back to (DANHIP_FORCE_DIST on one rank)")
        self.device_collectives = self.enabled and (self.fake or self.rccl is not None)
        # failure detection of the RCCL data plane (CommWatch): on whenever peers exist; DANHIP_COMM_WATCH=1 forces it for a one-rank
        # communicator (tests), =0 switches it off
        watch = os.environ.get("DANHIP_COMM_WATCH", "")
        self.watch = CommWatch(self.rccl) if (self.rccl is not None and watch != "0" and (self.rccl.world > 1 or watch == "1")) else None
        self.comm_stream = torch.cuda.Stream() if (self.enabled and self.on_gpu) else None     # (enable_local creates it for one-GPU runs)
        self.pending = []
        self.next_bucket = 0
        self.start_of = dict(getattr(flat, "start_of_member", None) or zip(flat.names, flat.starts))   # members of fused blocks map to their block
        self.done_upto = flat.total
        self.fired = set()
        # DANHIP_DP_CHECK=1 (tests): every bucket's reduced values are snapshotted on the communication stream and compared in finish()
        # with what the gradient buffer holds then — any difference is a gradient written AFTER its bucket had been reduced
        self.check = os.environ.get("DANHIP_DP_CHECK") == "1"
        self.snapshots = []
        # Two alternatives to the plain fp32 all-reduce, both device collectives only (RCCL), selected by environment:
        #   DANHIP_DP_COMM=rs_ag         reduce-scatter + all-gather of each bucket (the two halves of a ring all-reduce as separate
        #                                calls: the form a sharded optimizer would split between gradient and parameter traffic)
        #   DANHIP_DP_BUCKET_DTYPE=bf16  buckets travel as bf16 (half the xGMI bytes; each rank's gradient rounded once before the sum)
        self.comm = os.environ.get("DANHIP_DP_COMM", "allreduce")
        self.wire_bf16 = os.environ.get("DANHIP_DP_BUCKET_DTYPE", "f32") == "bf16"
        if self.comm not in ("allreduce", "rs_ag"):
            raise ValueError("DANHIP_DP_COMM must be allreduce or rs_ag")
        self.stage = None
        if self.enabled and (self.comm != "allreduce" or self.wire_bf16):
            if self.rccl is None:
                raise RuntimeError("DANHIP_DP_COMM / DANHIP_DP_BUCKET_DTYPE need device collectives (DANHIP_DP_TRANSPORT=rccl)")
            w = self.rccl.world
            longest = max(-(-(e - s) // w) * w for s, e in self.bounds)
            self.stage = [torch.zeros(longest, dtype=torch.bfloat16 if self.wire_bf16 else torch.float32, device=flat.g.device)
                          for _ in self.bounds]      # one staging buffer per bucket: several reductions are in flight at once

    @property
    def active(self):
        """the per-layer gradient-ready hooks matter: collectives to launch, or a per-bucket optimizer to run"""
        return self.enabled or self.local

    def enable_local(self):
        """One process, no collectives: keep the bucket machinery (gradient-ready events of both backward streams, the side stream) for the
        trainer's per-bucket optimizer."""
        if not self.enabled and self.on_gpu:
            self.local = True
            if self.comm_stream is None:
                self.comm_stream = torch.cuda.Stream()

    def _reduce(self, b, s, e):
        """Sum flat.g[s:e] over the ranks on the current (communication) stream; -> work handle or None."""
        g = self.flat.g[s:e]
        if self.rccl is None:                           # gloo (host-staged): a work handle the caller waits for
            return dist.all_reduce(g, op=dist.ReduceOp.SUM, async_op=True)
        if self.stage is None:
            self.rccl.all_reduce(g)                     # in place on the flat gradient buffer, asynchronous on the communication stream
            return None
        w, r = self.rccl.world, self.rccl.rank
        n = e - s
        padded = -(-n // w) * w
        st = self.stage[b][:padded]
        st[:n].copy_(g)                                 # (casts when the wire type is bf16; the padding tail stays zero)
        if self.comm == "rs_ag":
            per = padded // w
            shard = st[r * per:(r + 1) * per]
            self.rccl.reduce_scatter(shard, st)         # (in place: the shard is this rank's slice of the staging buffer)
            self.rccl.all_gather(st, shard)
        else:
            self.rccl.all_reduce(st)
        g.copy_(st[:n])
        return None

    def begin_step(self):
        self.next_bucket = 0
        self.pending = []
        self.done_upto = self.flat.total
        self.fired = set()
        self.snapshots = []

    def ready(self, name):
        """Gradient of `name` (and of everything created after it) is final."""
        if not self.active:
            return
        # The overlap rests on two properties of the step: a variable's gradient is produced by exactly ONE backward call, and the calls
        # arrive in reverse creation order (autograd runs nodes by decreasing sequence number; variables are created in forward order).
        # A variable consumed twice, or a hook arriving out of order, would be reduced before it is complete: refuse loudly.
        if name in self.fired:
            raise RuntimeError("GradBuckets.ready(%r) fired twice in one step: the variable is consumed by two ops, its first gradient "
                               "may already have been all-reduced" % name)
        self.fired.add(name)
        start = self.start_of[name]
        if start > self.done_upto:
            raise RuntimeError("GradBuckets.ready(%r): offset %d arrives after offsets >= %d were declared final (backward order is not "
                               "reverse creation order)" % (name, start, self.done_upto))
        self.done_upto = start
        self._launch_ready()

    def _launch_ready(self):
        while self.next_bucket < len(self.bounds) and self.bounds[self.next_bucket][0] >= self.done_upto:
            s, e = self.bounds[self.next_bucket]
            if self.on_gpu:
                from . import ops
                producers = {st.cuda_stream: st for st in [torch.cuda.current_stream()] + ops.wgrad_streams()}
                evs = []
                for st in producers.values():          # gradients are produced on the backward stream AND the weight-gradient stream
                    ev = torch.cuda.Event()
                    ev.record(st)
                    evs.append(ev)
                with torch.cuda.stream(self.comm_stream):
                    for ev in evs:
                        self.comm_stream.wait_event(ev)
                    if self.fake:
                        self.flat.g[s:e].mul_(1.0)
                    elif self.enabled:
                        wk = self._reduce(self.next_bucket, s, e)
                        if wk is not None:
                            self.pending.append(wk)
                            if self.check:
                                wk.wait()            # (orders the snapshot behind the collective on this stream; host-blocking for gloo)
                    if self.check:
                        self.snapshots.append((s, e, self.flat.g[s:e].clone()))
                    if self.on_bucket is not None:   # (device collectives only: the reduction above is stream-ordered in front of it)
                        self.on_bucket(self.next_bucket, s, e)
            else:                                    # gloo / CPU tensors (unit tests)
                wk = dist.all_reduce(self.flat.g[s:e], op=dist.ReduceOp.SUM, async_op=True)
                self.pending.append(wk)
                if self.check:
                    wk.wait()
                    self.snapshots.append((s, e, self.flat.g[s:e].clone()))
            self.next_bucket += 1

    def finish(self):
        if not self.active:
            return
        self.done_upto = 0
        self._launch_ready()
        for w in self.pending:
            w.wait()
        if self.on_gpu:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
            if self.watch is not None and not torch.cuda.is_current_stream_capturing():
                self.watch.tick(self.comm_stream)    # (a captured step is watched at replay: DetectorTrainer._graph_step)
        if self.check:
            from . import ops
            for st in ops.wgrad_streams():
                if st is not None:
                    torch.cuda.current_stream().wait_stream(st)
            for s, e, snap in self.snapshots:
                if not torch.equal(self.flat.g[s:e], snap):
                    bad = (self.flat.g[s:e] != snap).nonzero()[0].item() + s
                    names = [n for n, st0 in sorted(self.start_of.items(), key=lambda kv: kv[1]) if st0 <= bad]
                    raise RuntimeError("GradBuckets: the gradient at flat offset %d (variable %r) changed after its bucket [%d, %d) had been "
                                       "all-reduced" % (bad, names[-1] if names else "?", s, e))
            self.snapshots = []


def dp_transport():
    """'rccl' (default on a GPU: RcclComm, the library's own RCCL calls) or 'gloo' (the process group's host-staged collectives:
    DANHIP_DP_TRANSPORT=gloo — several ranks on ONE GPU in tests/test_zz_ddp_gpu.py — and every CPU run)."""
    t = os.environ.get("DANHIP_DP_TRANSPORT", "rccl" if torch.cuda.is_available() else "gloo").lower()
    if t not in ("rccl", "gloo"):
        raise ValueError("DANHIP_DP_TRANSPORT must be rccl or gloo, got %r" % t)
    return t


def init_distributed():
    """One process per GPU (torchrun).  Returns (rank, world, local_rank).

    The process group is the CONTROL plane (rendezvous, barriers, the unique id of the RCCL communicator, scalar statistics): backend
    gloo by default (DANHIP_DIST_BACKEND overrides), so no ProcessGroupNCCL watchdog / heartbeat thread exists in the process.  The
    gradients travel on RCCL through the library's own communicator (RcclComm / danhip_comm_*)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # (before the first HIP call of the process: the host driver only supports dmabuf IPC, and RCCL's peer mappings need this mode)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    # DANHIP_FORCE_DIST=1: run the bucketed all-reduce even for ONE rank — the only way to put the RCCL code path on a single-GPU box
    # (tests/test_zz_ddp_gpu.py); a one-rank job needs no process group for it
    if torch.cuda.is_available():
        local = local % torch.cuda.device_count()
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=os.environ.get("DANHIP_DIST_BACKEND", "gloo"), rank=rank, world_size=world)
    return rank, world, local


def shutdown_distributed(*trainers):
    """Orderly end of a data-parallel job: captured graphs (they hold the collectives' kernel nodes) first, then the streams drain, then
    the RCCL communicator, then the control-plane process group."""
    for tr in trainers:                              # (a dead peer would make the device waits below endless: the watchdog goes first)
        w = getattr(getattr(tr, "buckets", None), "watch", None)
        if w is not None:
            w.drain()
    for tr in trainers:
        if tr is not None:
            tr.close()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if RcclComm._shared is not None:
        RcclComm._shared.close()
    if dist.is_available() and dist.is_initialized():
        control_barrier("at shutdown")
        _control["group"], _control["made"] = None, False
        dist.destroy_process_group()
