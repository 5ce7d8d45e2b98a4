"""torch.autograd.Function wrappers over the libdanhip C ABI (device memory / streams / autograd are PyTorch
plumbing; all arithmetic happens in the HIP kernels).  Activations are NHWC 16-bit tensors (bf16, or fp16 with DANHIP_DTYPE=fp16: _lib.ACT_DTYPE).

Gradient sinks: when a parameter carries a `_danhip_grad` tensor (a view into the trainer's flat fp32 gradient
buffer) the backward kernels accumulate straight into it and return None to autograd; otherwise a fresh
gradient tensor is returned as usual (used by the parity tests with torch.autograd.grad).
"""
import contextlib
import ctypes
import os

import torch

from . import _lib
from ._lib import ACT_DTYPE as ACT, BF16, F32, SPLIT3, ConvDesc, call, ptr, stream


class OpsContext(object):
    """Everything mutable that STEERS the ops of this module, in one object (SURVEY 8b "no mutable globals on the data path"; VERDICT r4
    item 9): which kernel form an op launches, where it reports, and the per-step hooks a trainer arms.  Rounds 1-4 kept these as module
    attributes (`ops.USE_SLOTS = False` ...); those names still read and write through to the ACTIVE context (module __getattr__ /
    __setattr__ at the end of the file), so tests and A/B scripts are unchanged, but a caller can now hold its own:

        ctx = ops.OpsContext(USE_SPLITK=False)
        with ops.use_context(ctx): ...            # every op issued inside (and the backward of those ops, if run inside) sees ctx

    A DetectorTrainer keeps the context it was built under (`trainer.ops_ctx`) and runs its steps inside it.  The active context is
    process-wide, not thread-local: autograd runs backward functions on its own threads, and the deployment model is one process per GPU.
    What is NOT in here are caches (packed weights and their version stamp WEIGHT_EPOCH, scratch sizes): derived data, not steering.

    Kernel-form switches (results agree up to fp32 summation order whatever they say):
      USE_SPLITK       False: never pass the split-K scratch buffer (every shape on its single-pass kernel)
      USE_SLOTS        False: activation gradients travel through autograd's own edges instead of the direct hand-off (GradSlot)
      USE_RELU_BITS    [DANHIP_RELU_BITS, 1]  0: ReLU masks read as 16-bit activations instead of bit masks
      USE_POOL_ARG     [DANHIP_POOL_ARG, 1]   0: max-pool backward re-reads the activation instead of the 2-bit arg-max codes
      POOL_ONLY_TRAIN  [DANHIP_POOL_ONLY_TRAIN, 1]  0: the training forward of a conv whose only consumer is a fused pool still writes its map
      KEEP_DEFORM_COL  False: the deformable backward re-samples the im2col buffer as the reference does instead of keeping the forward's
      WGRAD_STREAM     [DANHIP_WGRAD_STREAM, 1]  0: weight gradients on the data gradients' stream
      WGRAD_FIRST      [DANHIP_WGRAD_FIRST, 0]   1: a convolution's backward issues its weight gradient (side stream) BEFORE its data gradient, so the
                       side stream waits for dY only, not for the data gradient.  Measured slower (round 6, same box, alternating: 12.81 against
                       12.73 ms): both kernels are full-chip persistent grids, "beside each other" means taking turns, and the data gradient —
                       the critical path — then queues behind the weight gradient's workgroups
      FUSE_FIRST_WGRAD [DANHIP_FUSE_FIRST_WGRAD, 1]  0: conv1_2's data gradient stores its output and conv1_1's weight gradient is a launch of its own
                       (rounds 1-5); 1: the first layer's weight / bias gradient is folded into the second layer's data gradient
                       (danhip_conv2d_bwd_data_bits_first: conv1_1's dY never reaches HBM) where a trainer's gradient sinks exist
      SPLIT_EVAL       True: convolutions of the fp32 inference path run as split-operand products on the fp16 MFMA (csrc/split_infer.hip;
                       models set it for precision "split"): fp32-accurate boxes at a third of the 16-bit rate instead of a tenth
    Diagnostic sinks (None = off): TRACE (tests: activations / decisions by variable id), PROFILE / PROFILE_BYTES (bench.py: HIP events
    and algorithmic bytes per convolution launch).
    Per-step state a trainer arms: GRAD_READY_HOOK (a parameter's gradient is final), LOSS_SCALE_DEV (device scalar of the dynamic loss
    scale), wgrad (the second backward stream: {"on", "side", "main", "keep"})."""
    __slots__ = ("USE_SPLITK", "USE_SLOTS", "USE_RELU_BITS", "USE_POOL_ARG", "POOL_ONLY_TRAIN", "KEEP_DEFORM_COL", "WGRAD_STREAM", "WGRAD_FIRST", "FUSE_FIRST_WGRAD", "SPLIT_EVAL", "TRACE",
                 "PROFILE", "PROFILE_BYTES", "GRAD_READY_HOOK", "LOSS_SCALE_DEV", "wgrad")

    def __init__(self, **overrides):
        env = os.environ.get
        self.USE_SPLITK = True
        self.USE_SLOTS = True
        self.USE_RELU_BITS = env("DANHIP_RELU_BITS", "1") == "1"
        self.USE_POOL_ARG = env("DANHIP_POOL_ARG", "1") == "1"
        self.POOL_ONLY_TRAIN = env("DANHIP_POOL_ONLY_TRAIN", "1") == "1"
        self.KEEP_DEFORM_COL = True
        self.WGRAD_STREAM = env("DANHIP_WGRAD_STREAM", "1") == "1"
        self.WGRAD_FIRST = env("DANHIP_WGRAD_FIRST", "0") == "1"
        self.FUSE_FIRST_WGRAD = env("DANHIP_FUSE_FIRST_WGRAD", "1") == "1"
        self.SPLIT_EVAL = False
        self.TRACE = self.PROFILE = self.PROFILE_BYTES = None
        self.GRAD_READY_HOOK = self.LOSS_SCALE_DEV = None
        self.wgrad = {"on": False, "side": None, "main": None, "keep": []}
        for k, v in overrides.items():
            setattr(self, k, v)                       # (AttributeError for a name that is not a field)


_CTX = OpsContext()                                   # the active context (rebound by use_context only)
_CTX_FIELDS = frozenset(OpsContext.__slots__) - {"wgrad"}


def context():
    """The active OpsContext."""
    return _CTX


@contextlib.contextmanager
def use_context(ctx):
    """Make `ctx` the active context for the duration of the block (process-wide: see OpsContext)."""
    global _CTX
    prev, _CTX = _CTX, ctx
    try:
        yield ctx
    finally:
        _CTX = prev



def _desc(N, H, W, Cin, Cout, kh, kw, stride, valid=False):
    d = ConvDesc()
    d.N, d.H, d.W, d.Cin, d.Cout, d.kh, d.kw, d.stride = N, H, W, Cin, Cout, kh, kw, stride
    if valid:                       # padding='valid': no padding, floor((in - k) / s) + 1 outputs
        d.Ho, d.Wo = (H - kh) // stride + 1, (W - kw) // stride + 1
    else:                           # padding='same' (TF): ceil(in / s) outputs, padding derived by the library
        d.Ho, d.Wo = -(-H // stride), -(-W // stride)
    return d


def packed_dims(d, which):
    r, c = ctypes.c_int64(), ctypes.c_int64()
    call("danhip_conv_packed_dims", ctypes.byref(d), which, ctypes.byref(r), ctypes.byref(c))
    return r.value, c.value


def pack_conv_weight(d, w_hwio, need_bwd=True):
    """fp32 HWIO [kh,kw,cin_real,Cout] -> (wf bf16 [Cout_pad,Kpad], wb bf16 [Cin_pad,Kpad_b] or None)."""
    assert w_hwio.dtype == torch.float32 and w_hwio.is_contiguous()
    rf, cf = packed_dims(d, 0)
    wf = torch.empty((rf, cf), dtype=ACT, device=w_hwio.device)
    wb = None
    if need_bwd:
        rb, cb = packed_dims(d, 1)
        wb = torch.empty((rb, cb), dtype=ACT, device=w_hwio.device)
    call("danhip_pack_conv_weight", ctypes.byref(d), ptr(w_hwio), w_hwio.shape[2], ptr(wf), ptr(wb), stream())
    return wf, wb


# ---- packed-weight cache.  A conv weight that is an nn.Parameter keeps its two bf16 packings in persistent buffers; they are
# reused while the parameter is unchanged (inference: no packing at all) and refreshed for ALL parameters by one launch
# after an optimizer update (repack_all).  "Unchanged" = same WEIGHT_EPOCH (bumped by the fused optimizer, which writes
# through raw pointers), same tensor version (in-place torch ops) and same storage.
WEIGHT_EPOCH = 0
_PACKED = {}
_PACK_TABLE_OF = {}          # key -> (entries, device table, workgroups, parameter ids, parameter pointers): repack_all(key, select);
                             # cleared whenever the set of cached packings or one of their buffers changes


class _Packed(object):
    __slots__ = ("ref", "key", "d", "wf", "wb", "stamp")


def _stamp(p):
    return (WEIGHT_EPOCH, p._version, p.data_ptr())


def _drop_packed(pid):
    _PACKED.pop(pid, None)
    _PACK_TABLE_OF.clear()


def packed_weights(d, w, w_param, need_bwd):
    """-> (wf, wb) for conv descriptor d; cached per Parameter, packed on the spot for plain tensors."""
    if w_param is None:
        return pack_conv_weight(d, w.detach(), need_bwd=need_bwd)
    import weakref
    key = (d.kh, d.kw, d.Cin, d.Cout, w.shape[2])
    pid = id(w_param)
    e = _PACKED.get(pid)
    if e is None or e.key != key or e.ref() is not w_param:
        e = _Packed()
        e.ref = weakref.ref(w_param, lambda _r, pid=pid: _drop_packed(pid))
        e.key, e.stamp, e.wb = key, None, None
        e.d = _desc(1, 8, 8, d.Cin, d.Cout, d.kh, d.kw, 1)           # packing depends on the kernel / channel dims only
        rf, cf = packed_dims(e.d, 0)
        e.wf = torch.empty((rf, cf), dtype=ACT, device=w.device)
        _PACKED[pid] = e
        _PACK_TABLE_OF.clear()
    if need_bwd and e.wb is None:
        rb, cb = packed_dims(e.d, 1)
        e.wb = torch.empty((rb, cb), dtype=ACT, device=w.device)
        e.stamp = None
        _PACK_TABLE_OF.clear()
    st = _stamp(w_param)
    if e.stamp != st:
        wd = w_param.detach()
        if not (wd.dtype == torch.float32 and wd.is_contiguous()):
            # e.g. a member of a fused block (VariableStore.fuse: a strided view of one flat-buffer segment once a trainer has laid the
            # block out) reached an UNFUSED convolution: the fused / unfused choice (net.danet FUSED_CONTEXT_BLOCK / FUSED_STAGE2_MIX,
            # DANHIP_FUSED_CONTEXT / DANHIP_FUSED_STAGE2_MIX) is per call, the layout is fixed when the trainer is built
            raise RuntimeError("packed_weights: a kernel variable of shape %s arrives as a non-contiguous view (strides %s): it is a member of a "
                               "fused parameter block, which only the fused op consumes - set the fused / unfused switches BEFORE the "
                               "trainer (FlatParams) is built, not between calls" % (tuple(wd.shape), tuple(wd.stride())))
        call("danhip_pack_conv_weight", ctypes.byref(e.d), ptr(wd), wd.shape[2], ptr(e.wf), ptr(e.wb), stream())
        e.stamp = st
    return e.wf, (e.wb if need_bwd else None)


def repack_all(key=None, select=None):
    """One launch refreshing every cached packing (call after the optimizer changed the parameters).
    key / select (round 5): a NAMED SUBSET of the cached packings - select(parameter) -> bool - with its own device table, for an optimizer
    that runs bucket by bucket while backward is still producing the other buckets' gradients (trainer.DetectorTrainer._bucket_opt)."""
    entries = [(e, e.ref()) for e in _PACKED.values()]
    entries = [(e, p) for e, p in entries if p is not None and (select is None or select(p))]
    if not entries:
        return
    tabs = _PACK_TABLE_OF
    tab = tabs.get(key)
    ids, ptrs = [id(p) for _, p in entries], [p.data_ptr() for _, p in entries]
    if tab is None or tab[3] != ids or tab[4] != ptrs:
        arr = (_lib.PackEntry * len(entries))()
        first = 0
        nb = ctypes.c_int32()
        for i, (e, p) in enumerate(entries):
            call("danhip_pack_entry_init", ctypes.byref(arr[i]), ctypes.byref(e.d), ptr(p.detach()), p.shape[2], ptr(e.wf), ptr(e.wb), first,
                 ctypes.byref(nb))
            first += nb.value
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8)
        tab = tabs[key] = (len(entries), host.to(entries[0][1].device), first, ids, ptrs)
    n, dev_tab, blocks, _, _ = tab
    call("danhip_pack_conv_weights_batched", ptr(dev_tab), n, blocks, stream())
    for e, p in entries:
        e.stamp = _stamp(p)


def _grad_sink(p):
    return getattr(p, "_danhip_grad", None)


def _sink_trainable(t):
    """A fused block (not a leaf itself) whose member parameters are being trained."""
    ms = getattr(t, "_danhip_members", None)
    return ms is not None and any(m.requires_grad for m in ms)


# Optional per-kernel timing (bench.py): when PROFILE is a dict, every conv forward / stride-1 data-gradient launch is
# bracketed by events on the launch stream and recorded under its kernel-instance label with its algorithmic FLOPs.
# (PROFILE: a field of OpsContext, see the top of the module)
# (PROFILE_BYTES: a field of OpsContext, see the top of the module)


def _prof_begin(st=None):
    if _CTX.PROFILE is None:
        return None
    e = torch.cuda.Event(enable_timing=True)
    e.record(st if st is not None else torch.cuda.current_stream())
    return e


def _prof_end(e0, d, which, st=None, wrote_y=True, pooled=False):
    """wrote_y = False: a pool-only launch (the full-resolution map is never stored); pooled: the launch also stores the 2x2-pooled map.
    Both only change the ALGORITHMIC byte count of the launch (conv1_2 / conv2_2 were credited the 839 / 419 MB map
    they no longer write, which filed an MFMA-bound launch under the HBM-bound ones)."""
    if e0 is None:
        return
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record(st if st is not None else torch.cuda.current_stream())
    if which == 2:
        label = _lib.lib().danhip_conv_wgrad_kernel_label(ctypes.byref(d)).decode()
    else:
        label = _lib.lib().danhip_conv_kernel_label(ctypes.byref(d), which).decode()
    cin = d.Cin                              # MACs are the same for fwd and dgrad: Ho*Wo*Cin*Cout*kh*kw per image
    flops = 2.0 * d.N * d.Ho * d.Wo * cin * d.Cout * d.kh * d.kw
    _CTX.PROFILE.setdefault(label, []).append((e0, e1, flops))
    if _CTX.PROFILE_BYTES is not None:            # algorithmic HBM bytes: activations in + out (16-bit; the heads write fp32), weights once
        co8 = (d.Cout + 7) // 8 * 8
        out_px = (d.Ho * d.Wo if wrote_y else 0) + (((d.Ho + 1) // 2) * ((d.Wo + 1) // 2) if pooled else 0)
        nbytes = 2.0 * d.N * (d.H * d.W * cin + out_px * co8) + 2.0 * d.kh * d.kw * cin * d.Cout
        _CTX.PROFILE_BYTES.setdefault(label, []).append(nbytes)


# ---- weight gradients on a second stream.  dgrad(L) and wgrad(L) both consume dY_L and are independent of each other; on one stream
# each kernel's ramp-up and tail (and, on the 40x40 / 20x20 maps, its under-filled grid) leave CUs idle that the other kernel can use
# (tools/probe_concurrent_bwd.py: conv4_2 0.95 -> 0.84 ms, conv5_1 0.37 -> 0.31, fc6 0.25 -> 0.18 for the pair).  The trainer switches
# this on around its backward pass and joins before the gradients are consumed; tensors the side stream still reads are kept alive
# until that join (no allocator reuse while in flight, also valid inside a hipGraph capture).
# (_WGRAD: a field of OpsContext, see the top of the module)
# read ONCE at import (DANHIP_WGRAD_STREAM=0: A/B on one stream); a process that wants to switch mid-run (bench.py's serialized
# roofline leg, tests) assigns ops.WGRAD_STREAM - the environment is never re-read or written
# (WGRAD_STREAM: a field of OpsContext, see the top of the module)


def wgrad_overlap_begin():
    if not torch.cuda.is_available():
        return
    if not _CTX.WGRAD_STREAM:
        return
    if _CTX.wgrad["side"] is None:
        _CTX.wgrad["side"] = torch.cuda.Stream()
    _CTX.wgrad["main"] = torch.cuda.current_stream()
    _CTX.wgrad["on"] = True


def wgrad_overlap_join():
    """The launching stream waits for every weight gradient issued on the side stream; releases the kept tensors."""
    if _CTX.wgrad["on"]:
        torch.cuda.current_stream().wait_stream(_CTX.wgrad["side"])
    _CTX.wgrad["on"] = False
    _CTX.wgrad["keep"].clear()


def wgrad_streams():
    """Streams gradient producers may be running on (the data-parallel buckets wait for all of them)."""
    return [_CTX.wgrad["main"], _CTX.wgrad["side"]] if _CTX.wgrad["on"] else []


# Optional callback(param) invoked right after a layer's weight/bias gradients have been produced in backward
# (the data-parallel trainer uses it to launch the bucketed gradient all-reduce while backward continues).
# (GRAD_READY_HOOK: a field of OpsContext, see the top of the module)



class GradSlot(object):
    """Direct gradient hand-off between this package's ops, bypassing autograd's per-edge tensors.

    Every activation produced here carries a slot.  In backward, a consumer that knows the slot writes (first delivery)
    or accumulates (later deliveries) its input-gradient contribution straight into the slot's buffer — already
    multiplied by (x > 0) when x is a ReLU output, which folds the producer's ReLU backward into the consumer's
    epilogue — and returns None to autograd.  The producer's backward then takes the buffer.  Consumers that do not
    know about slots (plain torch ops) still work: their gradient arrives through autograd and is merged in."""

    __slots__ = ("shape", "dtype", "device", "is_relu", "buf", "count")

    def __init__(self, t, is_relu, channels=None):
        self.shape, self.dtype, self.device, self.is_relu = t.shape, t.dtype, t.device, is_relu
        if channels is not None:                         # ragged Cout: the buffer carries the channel-padded gradient layout
            self.shape = tuple(t.shape[:-1]) + (channels,)
        self.buf, self.count = None, 0

    def target(self):
        """-> (buffer, accumulate flag) for the next delivery."""
        if self.buf is None:
            self.buf = torch.empty(self.shape, dtype=self.dtype, device=self.device)
            self.count = 0
        acc = 1 if self.count > 0 else 0
        self.count += 1
        return self.buf, acc

    def take(self):
        b, n = self.buf, self.count
        self.buf, self.count = None, 0
        return b if n > 0 else None


# USE_SLOTS = False (tests): every activation gradient travels through autograd's own edges instead of the direct hand-off, which gives
# the parity tests a second, independent route through the same kernels (tests/test_grad_parity_gpu.py)
# (USE_SLOTS: a field of OpsContext, see the top of the module)


def _slot_of(t):
    return getattr(t, "_dh_slot", None) if _CTX.USE_SLOTS else None


def _new_slot(track):
    return GradSlot.__new__(GradSlot) if (track and _CTX.USE_SLOTS) else None


def _attach_slot(t, is_relu):
    s = GradSlot(t, is_relu)
    t._dh_slot = s
    return s


# ReLU masks as bits for the data-gradient kernels that stage them in LDS (danhip_relu_bits).  One activation may feed several convolutions:
# every consumer's forward shares ONE holder hung on the activation tensor, the first backward that needs the bits fills it.
# (USE_RELU_BITS: a field of OpsContext, see the top of the module)


def _bits_holder(x):
    h = getattr(x, "_dh_bits", None)
    if h is None:
        h = [None]
        x._dh_bits = h
    return h


def relu_bits(x, holder):
    if holder[0] is None:
        C = x.shape[-1]
        M = x.numel() // C
        b = torch.empty((M, C // 8), dtype=torch.uint8, device=x.device)
        call("danhip_relu_bits", ptr(x), ptr(b), M, C, stream())
        holder[0] = b
    return holder[0]


# TRACE (tests): when a dict, every ReLU layer records its output under id(weight variable) and every 2x2 max-pool its input under
# "pools" (call order) — the discrete decisions of this forward pass, which the gradient-parity tests impose on the oracle graph
# (TRACE: a field of OpsContext, see the top of the module)

# (USE_SPLITK: a field of OpsContext, see the top of the module)


def _conv_scratch(d, which, dev):
    """Scratch buffer for a split-K launch of this forward (which = 0) / data-gradient (which = 1) call, or (None, 0): maps with too few
    output tiles to fill the chip (danhip_conv2d_workspace_bytes).  Only small problems are worth asking the library about."""
    M, co = (d.N * d.Ho * d.Wo, d.Cout) if which == 0 else (d.N * d.H * d.W, d.Cin)
    if not _CTX.USE_SPLITK or -(-M // 128) * -(-co // 128) > 160:
        return None, 0
    key = (which, d.N, d.H, d.W, d.Cin, d.Cout, d.kh, d.kw, d.stride, d.Ho)
    n = _SCRATCH_BYTES.get(key)
    if n is None:
        n = _SCRATCH_BYTES[key] = _lib.lib().danhip_conv2d_workspace_bytes(ctypes.byref(d), which)
    return (torch.empty(n, dtype=torch.uint8, device=dev), n) if n else (None, 0)


_SCRATCH_BYTES = {}          # (which, descriptor) -> bytes: one library query per shape (host time matters at 2 images per GPU)


def _wgrad_scratch(d, dev):
    key = (2, d.N, d.H, d.W, d.Cin, d.Cout, d.kh, d.kw, d.stride, d.Ho)
    n = _SCRATCH_BYTES.get(key)
    if n is None:
        n = _SCRATCH_BYTES[key] = _lib.lib().danhip_conv2d_bwd_weight_workspace_bytes(ctypes.byref(d))
    return (torch.empty(n, dtype=torch.uint8, device=dev), n) if n else (None, 0)


# USE_POOL_ARG (round 4): the 2 x 2 max-pool keeps 2-bit arg-max codes from its forward pass (written by the pooling conv epilogues / the pool
# kernel) and its backward scatters the pooled gradient through them instead of re-reading the full-resolution activation to find each
# window's maximum (danhip_maxpool2x2_bwd_arg: 1.28 instead of 2.25 map-sized HBM passes).  DANHIP_POOL_ARG=0: the round-3 form (A/B).
# (USE_POOL_ARG: a field of OpsContext, see the top of the module)


# DANHIP_POOL_ONLY_TRAIN=0: conv1_2 / conv2_2 write their full-resolution outputs in training too (A/B; round 4's behaviour)
# (POOL_ONLY_TRAIN: a field of OpsContext, see the top of the module)


def _pool_arg_buffer(pooled, need_bwd):
    if not (_CTX.USE_POOL_ARG and need_bwd):
        return None
    c = pooled.shape[-1]
    return torch.empty((pooled.numel() // c, c // 4), dtype=torch.uint8, device=pooled.device)


class _Conv2d(torch.autograd.Function):
    """y = act(conv2d_same(x, w) + b) [+ residual]; tf.layers.conv2d semantics (net/sfd_net.py:81-89)."""

    @staticmethod
    def forward(ctx, x, w, b, stride, relu, out_f32, residual, w_param, b_param, xslot, yslot, pool_out=None, valid=False, block_grads=0, xbits=None, bits_out=None,
                pool_only=False, tracked=True, first=None):
        N, H, W, C = x.shape
        kh, kw, cin_real, cout = w.shape
        ctx.first = first                                # (image, kernel, bias, real channels) of the FIRST layer when x is its output (conv2d)
        assert x.dtype == ACT and x.is_contiguous(), "conv input must be contiguous NHWC bf16"
        assert C % 8 == 0 and cin_real <= C
        d = _desc(N, H, W, C, cout, kh, kw, stride, valid)
        need_bwd = w.requires_grad or x.requires_grad or bool(block_grads)
        wf, wb = packed_weights(d, w, w_param, need_bwd)
        if pool_only == 1 and pool_out is not None and _lib.lib().danhip_conv2d_fwd_pool_only(ctypes.byref(d)):      # (nothing is tracked)
            # inference: nothing but the pool reads this activation - the kernel pools in its epilogue and never writes the full-resolution map
            pooled = torch.empty((N, (d.Ho + 1) // 2, (d.Wo + 1) // 2, cout), dtype=ACT, device=x.device)
            e0 = _prof_begin()
            call("danhip_conv2d_fwd_pool", ctypes.byref(d), ptr(x), ptr(wf), ptr(b.detach()), None, ptr(pooled), stream())
            _prof_end(e0, d, 4, wrote_y=False, pooled=True)
            pooled._dh_already_pooled = True
            return pooled
        # pool_only == 2 (training): nothing but the fused pool reads this activation, and backward reaches it only through the pool's arg-max
        # codes and the pooled map's sign (the gradient arrives already masked in the slot) - the kernel gets y = NULL and skips the
        # full-resolution stores (conv1_2: 839 MB, conv2_2: 419 MB per batch of 16 at 640 x 640).  The tensor autograd sees is a
        # zero-stride view of one element: any op that tried to READ it fails ptr()'s contiguity assertion instead of reading garbage.
        skip_y = (pool_only == 2 and pool_out is not None and _CTX.USE_POOL_ARG and need_bwd and b is not None
                  and bool(_lib.lib().danhip_conv2d_fwd_pool_only(ctypes.byref(d)))
                  and _conv_scratch(d, 0, x.device)[1] == 0)          # (a map small enough to split K runs the pool as its own kernel)
        if skip_y:
            y = torch.empty((1,), dtype=ACT, device=x.device).expand(N, d.Ho, d.Wo, cout)
        else:
            y = torch.empty((N, d.Ho, d.Wo, cout), dtype=torch.float32 if out_f32 else ACT, device=x.device)
        yp = None if skip_y else ptr(y)
        e0 = _prof_begin()
        emit = (bits_out is not None and relu and not out_f32 and residual is None and b is not None
                and _lib.lib().danhip_conv2d_fwd_emits_bits(ctypes.byref(d), 1 if pool_out is not None else 0))
        if emit:                                         # conv_relu (+ fused pool) that also leaves the ReLU bit masks for the next conv's data gradient
            ybits = torch.empty((N * d.Ho * d.Wo, cout // 8), dtype=torch.uint8, device=x.device)
            pooled = pbits = parg = None
            if pool_out is not None:
                pooled = torch.empty((N, (d.Ho + 1) // 2, (d.Wo + 1) // 2, cout), dtype=ACT, device=x.device)
                pbits = torch.empty((pooled.numel() // cout, cout // 8), dtype=torch.uint8, device=x.device)
                parg = _pool_arg_buffer(pooled, need_bwd and tracked)
                pool_out.extend([pooled, parg])
            call("danhip_conv2d_fwd_relu_bits_arg", ctypes.byref(d), ptr(x), ptr(wf), ptr(b.detach()), yp, ptr(ybits), ptr(pooled), ptr(pbits), ptr(parg),
                 stream())
            bits_out.extend([ybits, pbits])
        elif pool_out is not None:                       # conv_relu + the block's 2x2 max-pool in one call (fused epilogue where possible)
            assert relu and not out_f32 and residual is None and b is not None and cout % 8 == 0
            pooled = torch.empty((N, (d.Ho + 1) // 2, (d.Wo + 1) // 2, cout), dtype=ACT, device=x.device)
            parg = _pool_arg_buffer(pooled, need_bwd and tracked)
            ws, nws = _conv_scratch(d, 0, x.device)
            if nws:                                      # a map small enough to split K: no kernel of it fuses the pool anyway
                assert not skip_y
                call("danhip_conv2d_fwd_ws", ctypes.byref(d), ptr(x), ptr(wf), ptr(b.detach()), ptr(y), BF16, 1, None, ptr(ws), nws, stream())
                call("danhip_maxpool2x2_fwd_arg", ptr(y), ptr(pooled), ptr(parg), N, d.Ho, d.Wo, cout, stream())
            else:
                call("danhip_conv2d_fwd_pool_arg", ctypes.byref(d), ptr(x), ptr(wf), ptr(b.detach()), yp, ptr(pooled), ptr(parg), stream())
            pool_out.extend([pooled, parg])
        else:
            ws, nws = _conv_scratch(d, 0, x.device)
            call("danhip_conv2d_fwd_ws", ctypes.byref(d), ptr(x), ptr(wf), ptr(b.detach()) if b is not None else None, ptr(y),
                 F32 if out_f32 else BF16, int(relu), ptr(residual), ptr(ws), nws, stream())
        _prof_end(e0, d, 4 if pool_out is not None else 0, wrote_y=not skip_y, pooled=pool_out is not None)
        ctx.d, ctx.relu, ctx.cin_real = d, relu, cin_real
        ctx.xslot, ctx.yslot, ctx.xbits = xslot, yslot, xbits
        ctx.set_materialize_grads(False)
        ctx.has_res = residual is not None
        ctx.w_param, ctx.b_param = w_param, b_param
        ctx.save_for_backward(x, wb, y if (relu and not skip_y) else None)
        ctx.y_unwritten = skip_y
        ctx.has_bias = b is not None
        ctx.block_w, ctx.block_b = bool(block_grads & 1), bool(block_grads & 2)   # fused parameter blocks: not autograd leaves themselves
        return y

    @staticmethod
    def backward(ctx, dy):
        x, wb, y = ctx.saved_tensors
        d = ctx.d
        co8 = (d.Cout + 7) // 8 * 8
        M = d.N * d.Ho * d.Wo
        wp, bp = ctx.w_param, ctx.b_param
        need_dw = ctx.needs_input_grad[1] or ctx.block_w
        need_db = ctx.has_bias and (ctx.needs_input_grad[2] or ctx.block_b)
        db_sink = _grad_sink(bp) if bp is not None else None
        db = None
        if need_db:
            db = db_sink if db_sink is not None else torch.zeros(d.Cout, dtype=torch.float32, device=x.device)
        # ---- gather the output gradient: slot deliveries (already ReLU-masked) and/or the autograd tensor
        g = ctx.yslot.take() if ctx.yslot is not None else None
        dres = None
        if dy is not None and ctx.y_unwritten:
            raise RuntimeError("conv2d(pool_only=True): the full-resolution activation was never written, but a gradient reached it outside the "
                               "fused pool's direct hand-off (some op other than max_pool_2x2 consumed it)")
        if dy is not None:
            if ctx.has_res:
                dres = dy                                # residual is added after the activation
            if dy.dtype != ACT or dy.shape[-1] != co8:
                # fp32 / unpadded upstream gradient (head convs): cast + pad channels to a multiple of 8
                src = dy.contiguous().to(torch.float32)
                dyp = torch.empty((d.N, d.Ho, d.Wo, co8), dtype=ACT, device=dy.device)
                # ragged Cout: y has Cout (not co8) channels per row, so its ReLU mask is applied here, on the unpadded layout
                call("danhip_cast_pad_f32_to_bf16", ptr(src), ptr(y) if (ctx.relu and d.Cout != co8) else None, ptr(dyp), M, d.Cout, co8, stream())
                dy, owned = dyp, True
                masked = ctx.relu and d.Cout != co8
            else:
                dy, owned = dy.contiguous(), False
                masked = False
            if ctx.relu and not masked:
                if not owned:
                    dy = dy.clone()                      # the incoming gradient tensor may be shared with other consumers
                call("danhip_relu_bwd_bias_grad", ptr(dy), ptr(y), None, M, co8, stream())
            g = dy if g is None else g.add_(dy)
        if g is None:                                    # no gradient reached this layer
            return (None,) * 19
        db_in_wgrad = need_db and need_dw                        # the weight-gradient kernel also emits the bias gradient
        if need_db and not db_in_wgrad:
            if co8 == d.Cout:
                call("danhip_relu_bwd_bias_grad", ptr(g), None, ptr(db), M, co8, stream())
            else:
                db.add_(g.view(M, co8)[:, :d.Cout].to(torch.float32).sum(0))
        # Order of the two launches (ops.WGRAD_FIRST, round 6).  Default: data gradient first — the side stream's event is recorded BEHIND it, so
        # wgrad(L) waits for dgrad(L) to finish and runs beside dgrad(L-1); at the end of backward conv1_2's weight gradient (0.44 ms) and conv1_1's
        # (0.19 ms) run with the other queue empty (profiles/r5/s3fd_b16_step_timeline_full.txt: VERDICT r5's "0.63 ms single-queue tail").
        # WGRAD_FIRST issues the weight gradient first (its event then waits for dY only).  Measured on one box, alternating processes
        # (profiles/r6/ab_wgrad_first_same_box.txt): 12.81 ms against 12.73 for the default — SLOWER.  Two full-chip persistent grids cannot share
        # a CU, so "beside" means taking turns: what the second queue buys is tail filling, whatever the order, and with the weight gradient
        # ahead the data-gradient chain (the critical path: every later layer waits for it) queues behind 256 weight-gradient workgroups.  The
        # tail cannot be closed by reordering: conv1_2's two gradients are both chip-filling and conv1_1's needs conv1_2's data gradient.
        dx = None
        dw = None
        hooked = False
        fused_first = False

        def launch_dx():
            nonlocal dx, fused_first
            if ctx.needs_input_grad[0]:
                xs = ctx.xslot
                fi = ctx.first
                if (fi is not None and _CTX.FUSE_FIRST_WGRAD and xs is not None and xs.is_relu and xs.count == 0 and ctx.xbits is not None
                        and _grad_sink(fi[1]) is not None and (fi[2] is None or _grad_sink(fi[2]) is not None)
                        and _lib.lib().danhip_conv2d_bwd_data_first_supported(ctypes.byref(d))):
                    # x is the FIRST layer's output: that layer has no data gradient, so this call's dX would be read by nothing but its weight
                    # gradient — the kernel folds that product in (dX tile in LDS x image patch), adds it to the first layer's gradient sinks and
                    # never stores dX; the slot stays empty and the first layer's backward finds nothing to do
                    e0 = _prof_begin()
                    call("danhip_conv2d_bwd_data_bits_first", ctypes.byref(d), ptr(g), ptr(wb), ptr(relu_bits(x, ctx.xbits)), ptr(fi[0]), int(fi[3]),
                         ptr(_grad_sink(fi[1])), ptr(_grad_sink(fi[2])) if fi[2] is not None else None, stream())
                    _prof_end(e0, d, 5, wrote_y=False)       # (dX is never stored: algorithmic bytes = dY + bit mask + image + weights)
                    fused_first = True
                    return
                if xs is not None:                           # deliver straight into the producer's slot (+ its ReLU backward)
                    buf, acc = xs.target()
                    e0 = _prof_begin()
                    if xs.is_relu and ctx.xbits is not None and _lib.lib().danhip_conv2d_bwd_data_takes_bits(ctypes.byref(d)):
                        # the kernel keeps its tile's mask in LDS as bits: 1/16 of the bytes, and not a load in its epilogue
                        call("danhip_conv2d_bwd_data_bits", ctypes.byref(d), ptr(g), ptr(wb), ptr(relu_bits(x, ctx.xbits)), ptr(buf), acc, stream())
                    else:
                        ws, nws = _conv_scratch(d, 1, g.device)
                        call("danhip_conv2d_bwd_data_ws", ctypes.byref(d), ptr(g), ptr(wb), ptr(x) if xs.is_relu else None, ptr(buf), acc, ptr(ws), nws,
                             stream())
                    _prof_end(e0, d, 5 if xs.is_relu else 1)
                else:
                    dx = torch.empty_like(x)
                    e0 = _prof_begin()
                    ws, nws = _conv_scratch(d, 1, g.device)
                    call("danhip_conv2d_bwd_data_ws", ctypes.byref(d), ptr(g), ptr(wb), None, ptr(dx), 0, ptr(ws), nws, stream())
                    _prof_end(e0, d, 1)

        def launch_dw():
            nonlocal dw, hooked
            if need_dw:
                sink = _grad_sink(wp) if wp is not None else None
                dw = sink if sink is not None else torch.zeros((d.kh, d.kw, ctx.cin_real, d.Cout), dtype=torch.float32, device=g.device)
                # split partial sums as plain stores into a scratch slab + a combine pass, where the library's kernel for this shape offers it
                ws, nws = _wgrad_scratch(d, g.device)
                if _CTX.wgrad["on"] and sink is not None:      # side stream: needs dY (final now) and the zeroed sinks, both ordered on this stream
                    side = _CTX.wgrad["side"]
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream())
                    side.wait_event(ev)
                    e0 = _prof_begin(side)                  # (explicit stream handle: no stream-context switch per layer on the host)
                    call("danhip_conv2d_bwd_weight_ws", ctypes.byref(d), ptr(x), ptr(g), ptr(dw), ptr(db) if db_in_wgrad else None, ctx.cin_real,
                         ptr(ws), nws, ctypes.c_void_p(side.cuda_stream))
                    _prof_end(e0, d, 2, side)
                    if _CTX.GRAD_READY_HOOK is not None and wp is not None:
                        _CTX.GRAD_READY_HOOK(wp)                 # the buckets wait for both gradient streams (trainer.GradBuckets._launch_ready)
                    _CTX.wgrad["keep"].append((g, x, ws))
                    hooked = True
                else:
                    e0 = _prof_begin()
                    call("danhip_conv2d_bwd_weight_ws", ctypes.byref(d), ptr(x), ptr(g), ptr(dw), ptr(db) if db_in_wgrad else None, ctx.cin_real,
                         ptr(ws), nws, stream())
                    _prof_end(e0, d, 2)
                if sink is not None:
                    dw = None

        if _CTX.WGRAD_FIRST:
            launch_dw()
            launch_dx()
        else:
            launch_dx()
            launch_dw()
        if db_sink is not None:
            db = None
        if _CTX.GRAD_READY_HOOK is not None and wp is not None and not hooked:
            _CTX.GRAD_READY_HOOK(wp)
        if fused_first and _CTX.GRAD_READY_HOOK is not None:
            _CTX.GRAD_READY_HOOK(ctx.first[1])           # the first layer's gradients are final too (its own backward will find nothing to do)
        return dx, dw, db, None, None, None, dres, None, None, None, None, None, None, None, None, None, None, None, None


# ---- fp32 inference path (csrc/f32_infer.hip): every op below accepts fp32 NHWC activations and then runs the fp32 kernels — forward
# only (the evaluation graphs; "eval box outputs within 1e-4 of the reference").  The TF variables are used as they are (no packing).
def _f32_infer(x):
    if x.dtype != torch.float32:
        return False
    if torch.is_grad_enabled() and x.requires_grad:
        raise RuntimeError("the fp32 path is inference only (run it under torch.no_grad())")
    return True


def _conv2d_f32(x, w, b, stride, relu, residual, padding):
    N, H, W, C = x.shape
    kh, kw, cin, cout = w.shape
    assert cin == C and x.is_contiguous(), "fp32 conv: input channels must match the kernel (no channel padding on this path)"
    d = _desc(N, H, W, C, cout, kh, kw, stride, padding == "valid")
    y = torch.empty((N, d.Ho, d.Wo, cout), dtype=torch.float32, device=x.device)
    call("danhip_conv2d_fwd_f32", ctypes.byref(d), ptr(x), ptr(w.detach().contiguous()), ptr(b.detach()) if b is not None else None, ptr(y), int(relu),
         ptr(residual.contiguous()) if residual is not None else None, stream())
    return y


# ---- split-operand inference (csrc/split_infer.hip): the fp32 path's convolutions at the 16-bit MFMA rate.  fp32 NHWC tensors stay the
# currency between ops (every non-convolution op is the fp32 path's); a convolution takes its input in the 3C half layout [hi | lo | hi]
# — from the producing convolution's epilogue when it left one behind (`_dh_split3`), else converted here — and runs the fp16 build's
# ordinary kernels over 3C input channels with weights [hi | hi | lo].
_SPLIT_W = {}                # id(variable) -> (stamp, weakref, packed fp16 weights): evaluation weights are static, packed once


def _split_weight(d3, w, cin):
    """HWIO fp32 [kh, kw, cin, cout] -> the fp16 build's forward packing of [w_hi | w_hi | w_lo | 0] over d3.Cin = 3 cin (+ padding) channels."""
    import weakref
    key = id(w)
    stamp = (WEIGHT_EPOCH, w._version, w.data_ptr(), d3.Cin)
    e = _SPLIT_W.get(key)
    if e is not None and e[0] == stamp and e[1]() is w:
        return e[2]
    wd = w.detach().float()
    hi = wd.half().float()
    lo = (wd - hi).half().float()
    w3 = torch.cat([hi, hi, lo], dim=2).contiguous()
    r, c = ctypes.c_int64(), ctypes.c_int64()
    _lib.call_f16("danhip_conv_packed_dims", ctypes.byref(d3), 0, ctypes.byref(r), ctypes.byref(c))
    wf = torch.empty((r.value, c.value), dtype=torch.float16, device=w.device)
    _lib.call_f16("danhip_pack_conv_weight", ctypes.byref(d3), ptr(w3), 3 * cin, ptr(wf), None, stream())
    if isinstance(w, torch.nn.Parameter) or hasattr(w, "_danhip_grad"):        # (a per-call concatenation of head kernels is not worth caching)
        _SPLIT_W[key] = (stamp, weakref.ref(w, lambda _r, k=key: _SPLIT_W.pop(k, None)), wf)
    return wf


def _limb_view(x3, C):
    """The tensor a split-mode convolution hands on: shape [.., C] (the nets read channel counts from it), IEEE half, a strided view of the
    HI limbs of the [.., 3C] limb-layout map, which rides along as `_dh_split3`.  Consumers inside this module: conv2d and max_pool_2x2
    take the limb layout as it is; every other op widens it to fp32 first (_f32_in)."""
    v = x3[..., :C]
    v._dh_split3 = x3
    return v


def _is_limbs(x):
    return getattr(x, "_dh_split3", None) is not None


def unsplit3(x):
    """limb view -> contiguous fp32 NHWC (hi + lo)."""
    x3 = x._dh_split3
    C = x.shape[-1]
    y = torch.empty(tuple(x.shape), dtype=torch.float32, device=x.device)
    call("danhip_unsplit3_f32", ptr(x3), ptr(y), y.numel() // C, C, x3.shape[-1], stream())
    return y


def stop_gradient(x):
    """tf.stop_gradient.  A limb view (split-operand inference: nothing is differentiated) passes through — Tensor.detach() would return a
    plain half tensor without the limb map riding on it."""
    return x if _is_limbs(x) else x.detach()


def _f32_in(*ts):
    """Inputs of an op of the fp32 inference path: limb views (split-operand mode) widened to fp32, everything else untouched."""
    out = tuple(unsplit3(t) if (t is not None and _is_limbs(t)) else t for t in ts)
    return out[0] if len(out) == 1 else out


def split3(x):
    """fp32 NHWC [.., C] (or a limb view) -> IEEE-half [.., C3] = [hi | lo | hi | 0-padding], C3 = 3C rounded up to 8."""
    if _is_limbs(x):
        return x._dh_split3
    C = x.shape[-1]
    C3 = (3 * C + 7) // 8 * 8
    x3 = torch.empty(x.shape[:-1] + (C3,), dtype=torch.float16, device=x.device)
    call("danhip_split3_f32", ptr(x.contiguous()), ptr(x3), x.numel() // C, C, C3, 0, stream())
    return x3


def _conv2d_split(x, w, b, stride, relu, residual, padding, want_f32=False):
    """-> fp32 [N,Ho,Wo,cout] when the caller asks for it (heads), has a residual, or cout is ragged; else the next convolution's limb
    layout written by the kernel's epilogue (DANHIP_SPLIT3), returned as a limb view."""
    N, H, W, C = x.shape
    kh, kw, cin, cout = w.shape
    assert cin == C, "split conv: input channels must match the kernel"
    if (C == 3 and cout == 64 and kh == 3 and kw == 3 and stride == 1 and padding == "same" and residual is None and not want_f32
            and not _is_limbs(x) and x.dtype == torch.float32):
        # the first layer: 27 products per output - an exact fp32 FMA chain written straight in the limb layout (split_infer.hip)
        y3 = torch.empty((N, H, W, 3 * cout), dtype=torch.float16, device=x.device)
        call("danhip_conv3x3_c3_f32_split3", ptr(x.contiguous()), ptr(w.detach().float().contiguous()), ptr(b.detach().float()) if b is not None else None,
             ptr(y3), N, H, W, cout, int(relu), stream())
        return _limb_view(y3, cout)
    # tensors of the 16-bit kernels are addressed with 32-bit element offsets: 3C input channels (the deformable GEMM's 27 x 256 at 160 x 160)
    # or 3 Cout output limbs can exceed 2^31 elements at batch 16 — such a call runs in batch slices
    C3 = (3 * C + 7) // 8 * 8
    ho, wo = (((H - kh) // stride + 1), ((W - kw) // stride + 1)) if padding == "valid" else (-(-H // stride), -(-W // stride))
    nmax = ((1 << 31) - 1) // max(H * W * C3, ho * wo * 3 * ((cout + 7) // 8 * 8))
    if N > nmax >= 1:
        residual = _f32_in(residual)
        parts = [_conv2d_split(x[i:i + nmax] if not _is_limbs(x) else _limb_view(x._dh_split3[i:i + nmax], C), w, b, stride, relu,
                               None if residual is None else residual[i:i + nmax], padding, want_f32=True) for i in range(0, N, nmax)]
        return torch.cat(parts, dim=0)
    x3 = split3(x)
    d3 = _desc(N, H, W, x3.shape[-1], cout, kh, kw, stride, padding == "valid")
    wf = _split_weight(d3, w, cin)
    limbs = not want_f32 and residual is None and cout % 8 == 0 and N * d3.Ho * d3.Wo * 3 * cout < (1 << 31)
    if limbs:
        y = torch.empty((N, d3.Ho, d3.Wo, 3 * cout), dtype=torch.float16, device=x.device)
    else:
        y = torch.empty((N, d3.Ho, d3.Wo, cout), dtype=torch.float32, device=x.device)
    n = _lib.lib_f16().danhip_conv2d_workspace_bytes(ctypes.byref(d3), 0) if -(-(N * d3.Ho * d3.Wo) // 128) * -(-cout // 128) <= 160 else 0
    ws = torch.empty(n, dtype=torch.uint8, device=x.device) if n else None
    _lib.call_f16("danhip_conv2d_fwd_ws", ctypes.byref(d3), ptr(x3), ptr(wf), ptr(b.detach().float()) if b is not None else None, ptr(y),
                  SPLIT3 if limbs else F32, int(relu), None, ptr(ws), n, stream())
    if limbs:
        return _limb_view(y, cout)
    if residual is not None:                             # (added after the activation, as the 16-bit kernels' fused residual is)
        y.add_(_f32_in(residual))
    return y


def conv2d(x, w, b=None, stride=1, relu=False, out_f32=False, residual=None, pool=False, padding="same", pool_only=False):
    """pool=True: also computes max_pool_2x2(y) (danhip_conv2d_fwd_pool); the next ops.max_pool_2x2(y) call picks it up.
    pool_only=True (with pool, no gradient tracked): the caller promises that ONLY the pooled map is used - where the kernel pools in its
    epilogue the full-resolution activation is never written and the POOLED tensor is returned (ops.max_pool_2x2 passes it through)."""
    if _is_limbs(x) or (_CTX.SPLIT_EVAL and _f32_infer(x)):
        return _conv2d_split(x, w, b, stride, relu, residual, padding, want_f32=out_f32)
    if _f32_infer(x):
        return _conv2d_f32(x, w, b, stride, relu, _f32_in(residual), padding)
    # a plain tensor carrying a gradient sink is a fused block of parameters (FlatParams): cached packing, gradients written in place
    wp = w if (isinstance(w, torch.nn.Parameter) or hasattr(w, "_danhip_grad")) else None
    bp = b if (isinstance(b, torch.nn.Parameter) or hasattr(b, "_danhip_grad")) else None
    track = torch.is_grad_enabled() and (x.requires_grad or w.requires_grad or _sink_trainable(w))
    if track and relu and residual is not None:
        raise NotImplementedError("relu + fused residual needs a separate ReLU mask in backward (y > 0 is not the mask)")
    yslot = _new_slot(track and not out_f32)
    pool_out = [] if (pool and relu and not out_f32 and residual is None and b is not None and w.shape[-1] % 8 == 0) else None
    if padding not in ("same", "valid"):
        raise ValueError("padding must be 'same' or 'valid'")
    blk = 0
    if torch.is_grad_enabled():
        blk = (1 if _sink_trainable(w) else 0) | (2 if (b is not None and _sink_trainable(b)) else 0)
    xs = _slot_of(x) if track else None
    xbits = _bits_holder(x) if (xs is not None and xs.is_relu and _CTX.USE_RELU_BITS) else None
    bits_out = [] if (track and relu and _CTX.USE_RELU_BITS) else None
    # pool_only: 1 = nothing tracked, the pooled map is returned; 2 = training with the direct gradient hand-off: y is declared but never written
    # (POOL_ONLY_TRAIN; off while a test records activations through TRACE)
    po = (1 if not track else (2 if (_CTX.POOL_ONLY_TRAIN and _CTX.TRACE is None and _CTX.USE_SLOTS and yslot is not None) else 0)) if pool_only else 0
    y = _Conv2d.apply(x, w, b, stride, relu, out_f32, residual, wp, bp, xs, yslot, pool_out, padding == "valid", blk, xbits, bits_out,
                      po, bool(track), getattr(x, "_dh_first", None) if track else None)
    if (track and relu and not x.requires_grad and wp is not None and x.shape[-1] == 8 and w.shape[2] <= 4 and w.shape[3] == 64 and w.shape[0] == 3
            and w.shape[1] == 3 and stride == 1 and padding == "same" and residual is None and not out_f32):
        y._dh_first = (x, wp, bp, w.shape[2])               # the first layer (image in, no data gradient): its consumer may fold its weight gradient in
    if getattr(y, "_dh_already_pooled", False):
        return y
    if _CTX.TRACE is not None and relu and wp is not None:
        _CTX.TRACE[id(wp)] = y.detach()
    if bits_out:                                         # the forward kernel wrote the masks: the holders of y (and its pooled map) start filled
        y._dh_bits = [bits_out[0]]
    if yslot is not None:
        cout = w.shape[-1]
        if cout % 8 == 0:
            yslot.__init__(y, relu)
            y._dh_slot = yslot
        else:                                            # only ops.concat knows the padded layout (a ragged tensor feeds no conv directly)
            yslot.__init__(y, relu, (cout + 7) // 8 * 8)
            y._dh_pslot = yslot
    if pool_out:
        y._dh_pooled = pool_out[0]
        y._dh_pool_arg = pool_out[1]                     # 2-bit arg-max codes of the fused pool (None when nothing is tracked)
        if bits_out and bits_out[1] is not None:
            y._dh_pooled_bits = bits_out[1]
    return y


class _MaxPool(torch.autograd.Function):
    """tf.layers.max_pooling2d([2,2],[2,2],'same') — net/sfd_net.py:132."""

    @staticmethod
    def forward(ctx, x, xslot, yslot, pre=None, arg=None):
        N, H, W, C = x.shape
        if pre is not None:                              # already computed by the producing conv's epilogue (with its arg-max codes)
            y = pre
        else:
            y = torch.empty((N, (H + 1) // 2, (W + 1) // 2, C), dtype=x.dtype, device=x.device)
            arg = _pool_arg_buffer(y, x.requires_grad and C % 8 == 0)
            call("danhip_maxpool2x2_fwd_arg", ptr(x), ptr(y), ptr(arg), N, H, W, C, stream())
        ctx.arg = arg
        ctx.save_for_backward(x if arg is None else None)
        ctx.dims = (N, H, W, C)
        ctx.xslot, ctx.yslot = xslot, yslot
        ctx.set_materialize_grads(False)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        N, H, W, C = ctx.dims
        g = ctx.yslot.take() if ctx.yslot is not None else None
        if dy is not None:
            g = dy.contiguous() if g is None else g.add_(dy)
        if g is None:
            return None, None, None, None, None

        def scatter(dst, acc):
            if ctx.arg is not None:                      # through the forward pass's arg-max codes: the activation is not read again
                call("danhip_maxpool2x2_bwd_arg", ptr(ctx.arg), ptr(g), ptr(dst), N, H, W, C, acc, stream())
            else:
                call("danhip_maxpool2x2_bwd", ptr(x), ptr(g), ptr(dst), N, H, W, C, acc, stream())

        if ctx.xslot is not None:
            # the pooled maximum is > 0 exactly where its source is, so a gradient masked at the pooled level scatters to
            # an already ReLU-masked gradient; an unmasked one (autograd path) is masked by the producer's own backward
            buf, acc = ctx.xslot.target() if _pool_deliver_ok(ctx, dy) else (None, 0)
            if buf is not None:
                scatter(buf, acc)
                return None, None, None, None, None
        dx = torch.empty((N, H, W, C), dtype=g.dtype, device=g.device)
        scatter(dx, 0)
        return dx, None, None, None, None


def _pool_deliver_ok(ctx, dy):
    # direct delivery into a ReLU slot requires every contribution to be masked already, i.e. no raw autograd gradient
    return not (ctx.xslot.is_relu and dy is not None)


def max_pool_2x2(x):
    if getattr(x, "_dh_already_pooled", False):           # conv2d(pool=True, pool_only=True) already returned the pooled map
        return x
    if _is_limbs(x):                                      # split-operand mode: pooled straight on the limb layout
        N, H, W, C = x.shape
        if C % 8 == 0 and x._dh_split3.shape[-1] == 3 * C:
            y3 = torch.empty((N, (H + 1) // 2, (W + 1) // 2, 3 * C), dtype=torch.float16, device=x.device)
            call("danhip_maxpool2x2_split3", ptr(x._dh_split3), ptr(y3), N, H, W, C, stream())
            return _limb_view(y3, C)
        x = unsplit3(x)
    if _f32_infer(x):
        N, H, W, C = x.shape
        y = torch.empty((N, (H + 1) // 2, (W + 1) // 2, C), dtype=torch.float32, device=x.device)
        call("danhip_maxpool2x2_fwd_f32", ptr(x.contiguous()), ptr(y), N, H, W, C, stream())
        return y
    if _CTX.TRACE is not None:
        _CTX.TRACE.setdefault("pools", []).append(x.detach())
    track = torch.is_grad_enabled() and x.requires_grad
    xs = _slot_of(x) if track else None
    yslot = _new_slot(track)
    y = _MaxPool.apply(x, xs, yslot, getattr(x, "_dh_pooled", None), getattr(x, "_dh_pool_arg", None))
    pbits = getattr(x, "_dh_pooled_bits", None)
    if pbits is not None and getattr(x, "_dh_pooled", None) is not None:
        y._dh_bits = [pbits]                             # the fused conv + pool kernel also wrote the pooled map's ReLU bit mask
    if yslot is not None:
        # consumers may mask by (pooled > 0) when the source is a ReLU output
        yslot.__init__(y, xs.is_relu if xs is not None else False)
        y._dh_slot = yslot
    return y


class _MaxPool3x3S2(torch.autograd.Function):
    """tf.layers.max_pooling2d([3,3],[2,2],'same') — net/resnet_danet.py:129."""

    @staticmethod
    def forward(ctx, x):
        N, H, W, C = x.shape
        y = torch.empty((N, (H + 1) // 2, (W + 1) // 2, C), dtype=x.dtype, device=x.device)
        call("danhip_maxpool3x3s2_same_fwd", ptr(x), ptr(y), N, H, W, C, stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        N, H, W, C = x.shape
        dx = torch.empty_like(x)
        call("danhip_maxpool3x3s2_same_bwd", ptr(x), ptr(dy.contiguous()), ptr(dx), N, H, W, C, stream())
        return dx


def max_pool_3x3_s2(x):
    return _MaxPool3x3S2.apply(_f32_in(x))


class _L2Norm(torch.autograd.Function):
    """VGG16Backbone.l2_normalize — net/sfd_net.py:68-79."""

    @staticmethod
    def forward(ctx, x, gamma, g_param, xslot, yslot):
        y = torch.empty_like(x)
        M = x.numel() // x.shape[-1]
        call("danhip_l2norm_fwd", ptr(x), ptr(gamma.detach()), ptr(y), M, x.shape[-1], stream())
        ctx.save_for_backward(x, gamma.detach())
        ctx.g_param = g_param
        ctx.xslot, ctx.yslot = xslot, yslot
        ctx.set_materialize_grads(False)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma = ctx.saved_tensors
        M = x.numel() // x.shape[-1]
        g = ctx.yslot.take() if ctx.yslot is not None else None
        if dy is not None:
            g = dy.contiguous() if g is None else g.add_(dy)
        if g is None:
            return None, None, None, None, None
        sink = _grad_sink(ctx.g_param) if ctx.g_param is not None else None
        dg = sink if sink is not None else torch.zeros_like(gamma)
        xs = ctx.xslot
        if xs is not None:
            buf, acc = xs.target()
            call("danhip_l2norm_bwd", ptr(x), ptr(gamma), ptr(g), ptr(buf), ptr(dg), M, x.shape[-1], acc, 1 if xs.is_relu else 0, stream())
            dx = None
        else:
            dx = torch.empty_like(x)
            call("danhip_l2norm_bwd", ptr(x), ptr(gamma), ptr(g), ptr(dx), ptr(dg), M, x.shape[-1], 0, 0, stream())
        return dx, (None if sink is not None else dg), None, None, None


def l2_normalize(x, gamma):
    if _is_limbs(x) and x.shape[-1] % 8 == 0 and x._dh_split3.shape[-1] == 3 * x.shape[-1]:
        C = x.shape[-1]                                      # split-operand mode: normalised on the limb layout, for the head convolution that follows
        y3 = torch.empty_like(x._dh_split3)
        call("danhip_l2norm_split3", ptr(x._dh_split3), ptr(gamma.detach().float()), ptr(y3), y3.numel() // (3 * C), C, stream())
        return _limb_view(y3, C)
    x = _f32_in(x)
    if _f32_infer(x):
        y = torch.empty_like(x)
        call("danhip_l2norm_fwd_f32", ptr(x.contiguous()), ptr(gamma.detach()), ptr(y), x.numel() // x.shape[-1], x.shape[-1], stream())
        return y
    track = torch.is_grad_enabled() and (x.requires_grad or gamma.requires_grad)
    yslot = _new_slot(track)
    y = _L2Norm.apply(x, gamma, gamma if isinstance(gamma, torch.nn.Parameter) else None, _slot_of(x) if track else None, yslot)
    if yslot is not None:
        yslot.__init__(y, False)
        y._dh_slot = yslot
    return y


class _HeadSplit(torch.autograd.Function):
    """Max-out + reshape_pred for one pyramid level (depth 1): head conv output fp32 [B,H,W,4+nneg+npos] ->
    writes loc[B, off:off+HW, :], cls[B, off:off+HW, :] of the concatenated prediction buffers
    (net/sfd_net.py:175-216; train_sfd.py:293-304)."""

    @staticmethod
    def forward(ctx, h, loc, cls, nneg, npos, off):
        B, H, W, Ch = h.shape
        A = loc.shape[1]
        call("danhip_head_split_fwd", ptr(h), ptr(loc), ptr(cls), B, H * W, Ch, nneg, npos, A, off, stream())
        ctx.save_for_backward(h)
        ctx.cfg = (nneg, npos, off, A)
        ctx.mark_dirty(loc, cls)
        return loc, cls

    @staticmethod
    def backward(ctx, dloc, dcls):
        (h,) = ctx.saved_tensors
        nneg, npos, off, A = ctx.cfg
        B, H, W, Ch = h.shape
        dy = torch.empty((B, H, W, Ch), dtype=torch.float32, device=h.device)
        call("danhip_head_split_bwd", ptr(h), ptr(dloc.contiguous()), ptr(dcls.contiguous()), ptr(dy), B, H * W, Ch, nneg, npos, A, off, stream())
        return dy, dloc, dcls, None, None, None


def head_split(h, loc, cls, nneg, npos, off):
    return _HeadSplit.apply(h, loc, cls, nneg, npos, off)


# 1-element device tensor holding the current dynamic loss scale (set by the trainer around its backward pass), or None
# (LOSS_SCALE_DEV: a field of OpsContext, see the top of the module)


class _DetectionLoss(torch.autograd.Function):
    """Hard-negative mining + CE*(ratio+1) + smooth-L1 (train_sfd.py:350-417 / train_dan.py:286-324,470-478).
    Returns a 4-vector acc = [ce_sum, n_selected, loc_sum, n_pos] (device, no sync); the loss is
    (ratio+1)*ce_sum/n_selected + loc_sum/n_pos.  backward() expects the upstream gradient of that 4-vector to be
    ignored: it uses ctx.scale (= d total_loss / d this loss term) instead."""

    @staticmethod
    def forward(ctx, cls, loc, labels, loc_t, ratio, at_least_one, scale):
        B, A, _ = cls.shape
        # the kernels read raw pointers: a wrong dtype would be misread silently
        assert cls.dtype == torch.float32 and loc.dtype == torch.float32 and loc_t.dtype == torch.float32, "detection_loss: fp32 logits / targets"
        assert labels.dtype == torch.int32 and labels.shape == (B, A), "detection_loss: labels must be int32 [B, A]"
        assert cls.is_contiguous() and loc.is_contiguous() and labels.is_contiguous() and loc_t.is_contiguous()
        dev = cls.device
        score = torch.empty((B, A), dtype=torch.float32, device=dev)
        counts = torch.empty((B, 2), dtype=torch.int32, device=dev)
        thr = torch.empty((B,), dtype=torch.float32, device=dev)
        k = torch.empty((B,), dtype=torch.int32, device=dev)
        sel = torch.empty((B, A), dtype=torch.uint8, device=dev)
        acc = torch.empty((4,), dtype=torch.float32, device=dev)
        call("danhip_hard_neg_select", ptr(cls), ptr(labels), ptr(score), ptr(counts), ptr(thr), ptr(k), B, A, float(ratio), int(at_least_one), stream())
        call("danhip_detection_loss_fwd", ptr(cls), ptr(loc), ptr(labels), ptr(loc_t), ptr(score), ptr(thr), ptr(sel), ptr(acc), B, A, stream())
        if _CTX.TRACE is not None:                           # tests: the hard-negative selection of this term (call order), imposed on the oracle's loss
            _CTX.TRACE.setdefault("loss_sel", []).append(sel)
        ctx.save_for_backward(cls, loc, loc_t, sel, acc)
        ctx.cfg = (ratio, scale)
        ctx.aux = (score, thr, k, counts)
        return acc

    @staticmethod
    def backward(ctx, dacc):
        cls, loc, loc_t, sel, acc = ctx.saved_tensors
        ratio, scale = ctx.cfg
        B, A, _ = cls.shape
        dcls = torch.empty_like(cls)
        dloc = torch.empty_like(loc)
        call("danhip_detection_loss_bwd", ptr(cls), ptr(loc), ptr(loc_t), ptr(sel), ptr(acc), ptr(dcls), ptr(dloc), float((ratio + 1.0) * scale),
             float(scale), B, A, stream())
        if _CTX.LOSS_SCALE_DEV is not None:                  # dynamic loss scale (fp16 build): a device scalar, so the step stays capturable
            dcls.mul_(_CTX.LOSS_SCALE_DEV)
            dloc.mul_(_CTX.LOSS_SCALE_DEV)
        return dcls, dloc, None, None, None, None, None


def detection_loss(cls, loc, labels, loc_t, ratio=3.0, at_least_one=False, scale=1.0):
    return _DetectionLoss.apply(cls, loc, labels, loc_t, ratio, at_least_one, scale)


def preprocess_u8(img_rgb_u8):
    """uint8 RGB [N,H,W,3] -> bf16 [N,H,W,8] (BGR - mean, zero padded): dan_preprocessing.py:55-57,755-758."""
    assert img_rgb_u8.dtype == torch.uint8 and img_rgb_u8.shape[-1] == 3
    N, H, W, _ = img_rgb_u8.shape
    out = torch.empty((N, H, W, 8), dtype=ACT, device=img_rgb_u8.device)
    call("danhip_preprocess_u8", ptr(img_rgb_u8.contiguous()), ptr(out), N * H * W, stream())
    return out


def cast_pad(x_f32, c_dst):
    """fp32 [..., C] -> bf16 [..., c_dst] zero padded."""
    shp = x_f32.shape
    rows = x_f32.numel() // shp[-1]
    out = torch.empty(shp[:-1] + (c_dst,), dtype=ACT, device=x_f32.device)
    call("danhip_cast_pad_f32_to_bf16", ptr(x_f32.contiguous()), None, ptr(out), rows, shp[-1], c_dst, stream())
    return out


class _ResizeAdd(torch.autograd.Function):
    """out = lateral + tf.image.resize_bilinear(up, size(lateral)) (TF1 legacy mapping) — the LFPN merge of
    net/pb_net.py:209-217 / net/danet.py:363-371.  lateral may be None (plain resize to `size`)."""

    @staticmethod
    def forward(ctx, up, lateral, size, yslot=None):
        N, Hi, Wi, C = up.shape
        ctx.yslot = yslot
        ctx.set_materialize_grads(False)
        Ho, Wo = (lateral.shape[1], lateral.shape[2]) if lateral is not None else size
        assert up.dtype == ACT and up.is_contiguous() and C % 8 == 0
        out = torch.empty((N, Ho, Wo, C), dtype=ACT, device=up.device)
        call("danhip_resize_bilinear_add_fwd", ptr(up), ptr(lateral.contiguous()) if lateral is not None else None, ptr(out), N, Hi, Wi, Ho, Wo, C, stream())
        ctx.dims = (N, Hi, Wi, Ho, Wo, C)
        ctx.has_lat = lateral is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        N, Hi, Wi, Ho, Wo, C = ctx.dims
        # the merged map feeds two convolutions in the LFPN (this level's fused 3x3 and the next level's upsample 1x1): both deliver into its
        # slot (write, then accumulate in the kernel epilogue) instead of returning two tensors for the autograd engine to add
        g = ctx.yslot.take() if ctx.yslot is not None else None
        if dout is not None:
            g = dout.contiguous() if g is None else g.add_(dout)
        if g is None:
            return None, None, None, None
        dup = None
        if ctx.needs_input_grad[0]:
            dup = torch.empty((N, Hi, Wi, C), dtype=ACT, device=g.device)
            call("danhip_resize_bilinear_add_bwd", ptr(g), ptr(dup), N, Hi, Wi, Ho, Wo, C, 0, stream())
        return dup, (g if ctx.has_lat else None), None, None


def resize_bilinear_add(up, lateral=None, size=None):
    up, lateral = _f32_in(up, lateral)
    if _f32_infer(up):
        N, Hi, Wi, C = up.shape
        Ho, Wo = (lateral.shape[1], lateral.shape[2]) if lateral is not None else size
        out = torch.empty((N, Ho, Wo, C), dtype=torch.float32, device=up.device)
        call("danhip_resize_bilinear_add_fwd_f32", ptr(up.contiguous()), ptr(lateral.contiguous()) if lateral is not None else None, ptr(out), N, Hi, Wi,
             Ho, Wo, C, stream())
        return out
    track = torch.is_grad_enabled() and (up.requires_grad or (lateral is not None and lateral.requires_grad))
    yslot = _new_slot(track)
    out = _ResizeAdd.apply(up, lateral, size, yslot)
    if yslot is not None:
        yslot.__init__(out, False)
        out._dh_slot = yslot
    return out


class _AvgPool2x2S1(torch.autograd.Function):
    """tf.layers.average_pooling2d((2,2), 1, 'same') — net/danet.py:854."""

    @staticmethod
    def forward(ctx, x, xslot, yslot):
        N, H, W, C = x.shape
        y = torch.empty_like(x)
        call("danhip_avgpool2x2s1_same_fwd", ptr(x), ptr(y), N, H, W, C, stream())
        ctx.dims = (N, H, W, C)
        ctx.xslot, ctx.yslot = xslot, yslot
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x if (xslot is not None and xslot.is_relu) else None)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, H, W, C = ctx.dims
        g = ctx.yslot.take() if ctx.yslot is not None else None
        if dy is not None:
            g = dy.contiguous() if g is None else g.add_(dy)
        if g is None:
            return None, None, None
        xs = ctx.xslot
        if xs is not None:                               # deliver into the producer's slot (+ its ReLU backward)
            (x,) = ctx.saved_tensors
            buf, acc = xs.target()
            call("danhip_avgpool2x2s1_same_bwd", ptr(g), ptr(x), ptr(buf), N, H, W, C, acc, stream())
            return None, None, None
        dx = torch.empty((N, H, W, C), dtype=ACT, device=g.device)
        call("danhip_avgpool2x2s1_same_bwd", ptr(g), None, ptr(dx), N, H, W, C, 0, stream())
        return dx, None, None


def avg_pool_2x2_s1(x):
    x = _f32_in(x)
    if _f32_infer(x):
        N, H, W, C = x.shape
        y = torch.empty_like(x)
        call("danhip_avgpool2x2s1_same_fwd_f32", ptr(x.contiguous()), ptr(y), N, H, W, C, stream())
        return y
    assert x.dtype == ACT and x.is_contiguous() and x.shape[-1] % 8 == 0
    track = torch.is_grad_enabled() and x.requires_grad
    yslot = _new_slot(track)
    y = _AvgPool2x2S1.apply(x, _slot_of(x) if track else None, yslot)
    if yslot is not None:
        yslot.__init__(y, False)
        y._dh_slot = yslot
    return y


class _Concat(torch.autograd.Function):
    """tf.concat(values, axis=-1) (net/danet.py:911, :951; net/pb_net.py:306).  Backward hands each input its slice of dY directly
    (one pass: slice + the input's ReLU backward + accumulate), instead of autograd's strided views + .contiguous() + clone + mask."""

    @staticmethod
    def forward(ctx, slots, yslot, all_relu, *xs):
        y = torch.cat(xs, dim=-1)
        ctx.slots, ctx.yslot, ctx.all_relu = slots, yslot, all_relu
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(*[x if (s is not None and s.is_relu) else None for x, s in zip(xs, slots)])
        ctx.widths = [x.shape[-1] for x in xs]
        return y

    @staticmethod
    def backward(ctx, dy):
        g = ctx.yslot.take() if ctx.yslot is not None else None
        premasked = ctx.all_relu and dy is None          # slot deliveries arrive multiplied by (y > 0) = every input's own mask
        if dy is not None:
            dy = dy.contiguous()
            if g is not None and g.shape[-1] != dy.shape[-1]:             # ragged output width: the slot carries the channel-padded layout
                g[..., :dy.shape[-1]].add_(dy)
            else:
                g = dy if g is None else g.add_(dy)
        if g is None:
            return (None,) * (3 + len(ctx.widths))
        grads, c0 = [], 0
        for i, (C, s, x) in enumerate(zip(ctx.widths, ctx.slots, ctx.saved_tensors)):
            if not ctx.needs_input_grad[3 + i]:
                grads.append(None)
            elif s is None:
                grads.append(g[..., c0:c0 + C])
            else:
                buf, acc = s.target()
                call("danhip_slice_deliver", ptr(g), g.shape[-1], c0, C, ptr(x) if (s.is_relu and not premasked) else None, C, ptr(buf),
                     buf.shape[-1], acc, g.numel() // g.shape[-1], stream())
                grads.append(None)
            c0 += C
        return (None, None, None) + tuple(grads)


def concat(tensors):
    """Channel concatenation of NHWC activations."""
    tensors = [_f32_in(t) for t in tensors]
    track = torch.is_grad_enabled() and any(t.requires_grad for t in tensors)
    if tensors[0].dtype != ACT or not track or not _CTX.USE_SLOTS:
        return torch.cat(tensors, dim=-1)
    slots = [(getattr(t, "_dh_slot", None) or getattr(t, "_dh_pslot", None)) if t.requires_grad else None for t in tensors]
    all_relu = all(s is not None and s.is_relu for s in slots)
    yslot = _new_slot(True)
    y = _Concat.apply(slots, yslot, all_relu, *tensors)
    if y.shape[-1] % 8 == 0:
        yslot.__init__(y, all_relu)
        y._dh_slot = yslot
    else:                                                # ragged width: consumers deliver into the channel-padded layout (Cpad % 8 == 0)
        yslot.__init__(y, all_relu, (y.shape[-1] + 7) // 8 * 8)
        y._dh_pslot = yslot
    return y


class _Add(torch.autograd.Function):
    """a + b of two activations (the context module's residual, net/danet.py:913-918 / net/danet_deform.py:286-290): one library kernel forward
    (danhip_add16); backward delivers dY into both producers' slots, each with its own ReLU backward folded in — in ONE pass over dY
    (danhip_residual_bwd) when `a` is a ReLU output receiving its first delivery (the shape of every residual in the reference graphs)."""

    @staticmethod
    def forward(ctx, a, b, sa, sb, yslot):
        ctx.slots, ctx.yslot = (sa, sb), yslot
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(*[t if (s is not None and s.is_relu) else None for t, s in ((a, sa), (b, sb))])
        if a.dtype == ACT and a.is_contiguous() and b.is_contiguous() and a.is_cuda and a.numel() % 8 == 0:
            out = torch.empty_like(a)
            call("danhip_add16", ptr(a), ptr(b), ptr(out), a.numel(), stream())
            return out
        return a + b

    @staticmethod
    def backward(ctx, dy):
        g = ctx.yslot.take() if ctx.yslot is not None else None
        if dy is not None:
            g = dy.contiguous() if g is None else g.add_(dy)
        if g is None:
            return (None,) * 5
        sa, sb = ctx.slots
        ta, tb = ctx.saved_tensors
        if (sa is not None and sb is not None and sa.is_relu and ctx.needs_input_grad[0] and ctx.needs_input_grad[1] and sa.buf is None
                and g.numel() % 8 == 0):
            bufa, _ = sa.target()                                # first delivery into a's slot: written, not accumulated
            bufb, accb = sb.target()
            call("danhip_residual_bwd", ptr(g), ptr(ta), ptr(tb) if sb.is_relu else None, ptr(bufa), ptr(bufb), accb, g.numel(), stream())
            return None, None, None, None, None
        out = []
        for i, (s, t) in enumerate(zip(ctx.slots, ctx.saved_tensors)):
            if not ctx.needs_input_grad[i]:
                out.append(None)
            elif s is None:
                out.append(g)
            else:
                buf, acc = s.target()
                C = g.shape[-1]
                call("danhip_slice_deliver", ptr(g), C, 0, C, ptr(t) if s.is_relu else None, C, ptr(buf), C, acc, g.numel() // C, stream())
                out.append(None)
        return out[0], out[1], None, None, None


def add(a, b):
    a, b = _f32_in(a, b)
    track = torch.is_grad_enabled() and (a.requires_grad or b.requires_grad)
    if a.dtype != ACT or not track or not _CTX.USE_SLOTS or a.shape != b.shape or a.shape[-1] % 8:
        return a + b
    yslot = _new_slot(True)
    y = _Add.apply(a, b, _slot_of(a) if a.requires_grad else None, _slot_of(b) if b.requires_grad else None, yslot)
    yslot.__init__(y, False)
    y._dh_slot = yslot
    return y


def _vptr(t):
    """Device pointer of an NHWC activation that may be a CHANNEL-SLICE VIEW of a wider tensor (dense channel axis, pixels t.stride(-2)
    elements apart): what the *_strided entry points take together with that pitch."""
    assert t.dim() == 4 and t.stride(3) == 1, "channel-slice view: NHWC with a dense channel axis"
    N, H, W, _ = t.shape
    ld = t.stride(2)
    assert ld % 8 == 0 and t.stride(1) == W * ld and t.stride(0) == H * W * ld and t.storage_offset() % 8 == 0, "not a channel slice of a contiguous NHWC tensor"
    return ctypes.c_void_p(t.data_ptr())


def _wgrad_launch(d, xv, dyv, dw, db, cin_real, keep):
    """Weight (+ bias) gradient of descriptor d with x / dy given as (possibly channel-slice) views, on the weight-gradient stream when the
    trainer has it on (same ordering rules as _Conv2d.backward)."""
    pitch = _lib.ConvPitch(xv.stride(2), dyv.stride(2), 0)
    ws, nws = _wgrad_scratch(d, dyv.device)
    if _CTX.wgrad["on"]:
        side = _CTX.wgrad["side"]
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        side.wait_event(ev)
        e0 = _prof_begin(side)
        call("danhip_conv2d_bwd_weight_strided", ctypes.byref(d), _vptr(xv), _vptr(dyv), ptr(dw), ptr(db), cin_real, ctypes.byref(pitch), ptr(ws), nws,
             ctypes.c_void_p(side.cuda_stream))
        _prof_end(e0, d, 2, side)
        _CTX.wgrad["keep"].append((xv, dyv, ws) + tuple(keep))
    else:
        e0 = _prof_begin()
        call("danhip_conv2d_bwd_weight_strided", ctypes.byref(d), _vptr(xv), _vptr(dyv), ptr(dw), ptr(db), cin_real, ctypes.byref(pitch), ptr(ws), nws, stream())
        _prof_end(e0, d, 2)


class _ContextBlock(torch.autograd.Function):
    """DAN context module V1, se_inception_block (net/danet.py:842-918), as ONE autograd node over channel-slice views — round 4.

    Forward (8 launches instead of 13 + torch.cat + torch add):
        hyper[.., 0:64]      = relu(conv1x1_b1(x))                                   written straight into the concat buffer
        T = [b3 | b4 | p2]   = conv1x1(x) with the three kernels side by side (192 columns; ReLU on the first 128 only)
        hyper[.., 64:128]    = relu(avg_pool_2x2_s1(p2))        branch 2's 1x1 commuted in front of the (linear) pool: conv(avg(x)) = avg(conv(x)),
                                                                its bias included (the average of a constant is the constant)
        hyper[.., 128:192]   = relu(conv3x1(b3)) | relu(conv1x3(b3))                 ONE 3x3 convolution 64 -> 64 whose kernel holds the 3x1 taps in its
                                                                middle column for outputs 0..31 and the 1x3 taps in its middle row for outputs 32..63,
                                                                zeros elsewhere ("plus" block of the flat buffers): the register-resident 64 -> 64 kernel
                                                                (conv_halo_c64.hip) and the row-streaming weight gradient take it on the large levels
        U                    = relu(conv3x3(b4));  hyper[.., 192:256] = relu(conv3x1(U)) | relu(conv1x3(U))       (the same way)
        out                  = relu(conv1x1_res(hyper)) + x
    Backward: d hyper comes out of the residual conv's data gradient already multiplied by (hyper > 0) = every branch's own ReLU mask;
    each branch convolution reads ITS slice of it in place (no slice copies), gradients meet in dT / dU by accumulation, and x receives
    three deliveries (skip path, fused 1x1, b1) instead of seven.  Variable names, creation order and arithmetic per output element are
    the reference's."""

    @staticmethod
    def forward(ctx, x, xslot, yslot, handles, hook_order, *wb):
        N, H, W, C = x.shape
        dev = x.device
        assert x.dtype == ACT and x.is_contiguous() and C % 64 == 0
        pairs = list(zip(wb[0::2], wb[1::2]))                       # b1, cat, b3 plus (3x1 | 1x3), b43, b4 plus, res
        need_bwd = any(t.requires_grad for t in wb) or x.requires_grad or any(_sink_trainable(t) for t in wb)
        T = torch.empty((N, H, W, 192), dtype=ACT, device=dev)
        hyper = torch.empty((N, H, W, 256), dtype=ACT, device=dev)
        U = torch.empty((N, H, W, 64), dtype=ACT, device=dev)
        r = torch.empty((N, H, W, C), dtype=ACT, device=dev)
        out = torch.empty((N, H, W, C), dtype=ACT, device=dev)
        descs, wbs = [], []

        def conv(i, xv, yv, kh, kw, relu_ch):
            w, b = pairs[i]
            d = _desc(N, H, W, xv.shape[-1], yv.shape[-1], kh, kw, 1)
            wf, wbk = packed_weights(d, w, handles[i][0], need_bwd)
            pitch = _lib.ConvPitch(xv.stride(2), yv.stride(2), 0)
            sc, nsc = _conv_scratch(d, 0, dev)
            e0 = _prof_begin()
            call("danhip_conv2d_fwd_strided", ctypes.byref(d), _vptr(xv), ptr(wf), ptr(b.detach()), _vptr(yv), 1, relu_ch, ctypes.byref(pitch), ptr(sc), nsc,
                 stream())
            _prof_end(e0, d, 0)
            descs.append(d)
            wbs.append(wbk)

        conv(0, x, hyper[..., 0:64], 1, 1, 64)
        conv(1, x, T, 1, 1, 128)
        call("danhip_avgpool2x2s1_same_fwd_strided", _vptr(T[..., 128:192]), 192, _vptr(hyper[..., 64:128]), 256, N, H, W, 64, 1, stream())
        conv(2, T[..., 0:64], hyper[..., 128:192], 3, 3, 64)
        conv(3, T[..., 64:128], U, 3, 3, 64)
        conv(4, U, hyper[..., 192:256], 3, 3, 64)
        conv(5, hyper, r, 1, 1, C)
        call("danhip_add16", ptr(r), ptr(x), ptr(out), out.numel(), stream())
        ctx.descs, ctx.handles, ctx.hook_order = descs, handles, hook_order
        ctx.xslot, ctx.yslot = xslot, yslot
        ctx.nw = len(wb)
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, T, hyper, U, r, *wbs)
        ctx.blocks = [(_sink_trainable(w), _sink_trainable(b)) for w, b in pairs]
        ctx.cin_real = [w.shape[2] for w, _ in pairs]
        ctx.couts = [w.shape[3] for w, _ in pairs]
        ctx.khw = [(w.shape[0], w.shape[1]) for w, _ in pairs]
        if _CTX.TRACE is not None:                           # the block's ten ReLU decisions (tests): context_block() files them under their kernel variables
            _CB_LAST.clear()
            _CB_LAST.update({"b1": hyper[..., 0:64], "b2": hyper[..., 64:128], "b3": T[..., 0:64], "b3a": hyper[..., 128:160], "b3b": hyper[..., 160:192],
                             "b4": T[..., 64:128], "b43": U, "b4a": hyper[..., 192:224], "b4b": hyper[..., 224:256], "res": r})
        return out

    @staticmethod
    def backward(ctx, dy):
        x, T, hyper, U, r = ctx.saved_tensors[:5]
        wbs = ctx.saved_tensors[5:]
        N, H, W, C = x.shape
        dev = x.device
        g = ctx.yslot.take() if ctx.yslot is not None else None
        if dy is not None:
            g = dy.contiguous() if g is None else g.add_(dy)
        if g is None:
            return (None,) * (5 + ctx.nw)
        xs = ctx.xslot
        need_dx = ctx.needs_input_grad[0]
        dx_ret, xbuf, xacc, xmask = None, None, 0, None
        if need_dx:
            if xs is not None:
                xbuf, xacc = xs.target()
                xmask = x if xs.is_relu else None
            else:
                xbuf = dx_ret = torch.empty_like(x)
        gr = torch.empty_like(g)
        call("danhip_residual_bwd", ptr(g), ptr(r), ptr(xmask), ptr(gr), ptr(xbuf), xacc, g.numel(), stream())
        dhyper = torch.empty_like(hyper)
        dT = torch.empty_like(T)
        dU = torch.empty_like(U)
        grads = [None] * ctx.nw

        def sinks(i):
            wp, bp = ctx.handles[i]
            need_dw = ctx.needs_input_grad[5 + 2 * i] or ctx.blocks[i][0]
            need_db = ctx.needs_input_grad[6 + 2 * i] or ctx.blocks[i][1]
            dw = db = None
            if need_dw:
                sk = _grad_sink(wp) if wp is not None else None
                dw = sk if sk is not None else torch.zeros(ctx.khw[i] + (ctx.cin_real[i], ctx.couts[i]), dtype=torch.float32, device=dev)
                if sk is None:
                    grads[2 * i] = dw
            if need_db:
                sk = _grad_sink(bp) if bp is not None else None
                db = sk if sk is not None else torch.zeros(ctx.couts[i], dtype=torch.float32, device=dev)
                if sk is None:
                    grads[2 * i + 1] = db
            return dw, db

        def dgrad(i, dyv, mask, dxv, acc):
            d = ctx.descs[i]
            pitch = _lib.ConvPitch(dyv.stride(2), dxv.stride(2), mask.stride(2) if mask is not None else 0)
            sc, nsc = _conv_scratch(d, 1, dev)
            e0 = _prof_begin()
            call("danhip_conv2d_bwd_data_strided", ctypes.byref(d), _vptr(dyv), ptr(wbs[i]), _vptr(mask) if mask is not None else None, _vptr(dxv), acc,
                 ctypes.byref(pitch), ptr(sc), nsc, stream())
            _prof_end(e0, d, 5 if mask is not None else 1)

        def wgrad(i, xv, dyv):
            dw, db = sinks(i)
            if dw is None:
                if db is not None:                               # (never in the reference graphs: a trainable bias comes with its kernel)
                    db.add_(dyv.to(torch.float32).sum((0, 1, 2)))
                return
            _wgrad_launch(ctx.descs[i], xv, dyv, dw, db, ctx.cin_real[i], (gr, dhyper, dT, dU))

        # residual conv: d hyper = (gr . W^T) * (hyper > 0); then every branch on its slice.  (The weight gradients of the two "plus" kernels
        # also fill the taps / columns their zeros occupy: FlatParams.mask_structured clears those before the optimizer, plain autograd
        # slices them away.)
        dgrad(5, gr, hyper, dhyper, 0)
        wgrad(5, hyper, gr)
        dgrad(4, dhyper[..., 192:256], U, dU, 0)
        wgrad(4, U, dhyper[..., 192:256])
        dgrad(3, dU, T[..., 64:128], dT[..., 64:128], 0)
        wgrad(3, T[..., 64:128], dU)
        dgrad(2, dhyper[..., 128:192], T[..., 0:64], dT[..., 0:64], 0)
        wgrad(2, T[..., 0:64], dhyper[..., 128:192])
        call("danhip_avgpool2x2s1_same_bwd_strided", _vptr(dhyper[..., 64:128]), 256, _vptr(dT[..., 128:192]), 192, N, H, W, 64, stream())
        if need_dx:
            dgrad(1, dT, xmask, xbuf, 1)
        wgrad(1, x, dT)
        if need_dx:
            dgrad(0, dhyper[..., 0:64], xmask, xbuf, 1)
        wgrad(0, x, dhyper[..., 0:64])
        if _CTX.GRAD_READY_HOOK is not None:
            for p_ in ctx.hook_order:                            # all of the block's gradients are issued: declare them in reverse creation order
                _CTX.GRAD_READY_HOOK(p_)
        return (dx_ret, None, None, None, None) + tuple(grads)


_CB_LAST = {}


def context_block(x, params, hook_order, trace_params=None):
    """params: six (w, b) pairs in the order b1, cat(b3 | b4 | b2), plus(b3a 3x1 | b3b 1x3) as a [3, 3, 64, 64] kernel, b43 (3x3), plus(b4a | b4b), res;
    hook_order: the block's kernel Parameters in reverse creation order (data-parallel gradient buckets); trace_params: {"b1", "b2", "b3",
    "b3a", "b3b", "b4", "b43", "b4a", "b4b", "res"} -> kernel Parameter (ops.TRACE: tests).  -> out (with a gradient slot)."""
    assert not _is_limbs(x), "context_block is a 16-bit training op (the nets take the unfused block on the fp32 / split paths)"
    track = torch.is_grad_enabled()
    handles, flat = [], []
    for w, b in params:
        wp = w if (isinstance(w, torch.nn.Parameter) or hasattr(w, "_danhip_grad")) else None
        bp = b if (isinstance(b, torch.nn.Parameter) or hasattr(b, "_danhip_grad")) else None
        handles.append((wp, bp))
        flat += [w, b]
    yslot = _new_slot(track)
    out = _ContextBlock.apply(x, _slot_of(x) if (track and x.requires_grad) else None, yslot, handles, hook_order, *flat)
    if _CTX.TRACE is not None and trace_params:
        for k, prm in trace_params.items():
            _CTX.TRACE[id(prm)] = _CB_LAST[k].detach().contiguous()
        _CB_LAST.clear()
    if yslot is not None:
        yslot.__init__(out, False)
        out._dh_slot = yslot
    return out


class _ConcatMix(torch.autograd.Function):
    """out = relu([a | f] . Wv + bv) with a BLOCK-DIAGONAL Wv = diag(W1, W2): two 1x1 convolutions of two different maps written side by side
    into one map, i.e. concat([relu(conv1x1(a, W1)), relu(conv1x1(f, W2))]) — DAN's stage-2 input mix (net/danet.py:944-950), whose
    85 / 171 (170 / 342, 341 / 683) column counts fit no tile and went through the flat-M kernel plus torch.cat and, in backward, a slice
    pass per part — as ONE streaming GEMM over the never-materialised concatenation (danhip_conv2d_fwd_concat2) — round 4.
    `a` carries no gradient (the reference puts a stop_gradient on it).  Backward: the weight gradient of each input's [C, Co] row block by the
    ordinary 1x1 kernel against the FULL output gradient, after which the off-diagonal blocks (columns of the other part) are cleared;
    f's data gradient through the row block of f over all output columns (zeros where W1's columns are): dense 256-wide tiles, no ragged K."""

    @staticmethod
    def forward(ctx, a, f, fslot, yslot, handles, split, wv, bv):
        N, H, W, C1 = a.shape
        C2 = f.shape[-1]
        Co = wv.shape[3]
        dev = a.device
        assert a.dtype == ACT and f.dtype == ACT and a.is_contiguous() and f.is_contiguous() and a.shape[:3] == f.shape[:3]
        assert tuple(wv.shape[:3]) == (1, 1, C1 + C2) and wv.is_contiguous() and not a.requires_grad
        wvp, bvp, lowp = handles
        need_bwd = f.requires_grad or wv.requires_grad or bv.requires_grad or _sink_trainable(wv)
        d = _desc(N, H, W, C1 + C2, Co, 1, 1, 1)
        wf, _ = packed_weights(d, wv, wvp, False)
        out = torch.empty((N, H, W, Co), dtype=ACT, device=dev)
        e0 = _prof_begin()
        if C1 == C2 and _lib.lib().danhip_conv2d_fwd_concat2_supported(ctypes.byref(d), C1, C1):
            call("danhip_conv2d_fwd_concat2", ctypes.byref(d), ptr(a), ptr(f), C1, C1, ptr(wf), ptr(bv.detach()), ptr(out), 1, stream())
        else:                                            # small maps (the 20^2 .. 5^2 levels): concatenate, then the split-K / flat-M kernels
            x = torch.cat([a, f], dim=-1)
            ws, nws = _conv_scratch(d, 0, dev)
            call("danhip_conv2d_fwd_ws", ctypes.byref(d), ptr(x), ptr(wf), ptr(bv.detach()), ptr(out), BF16, 1, None, ptr(ws), nws, stream())
        _prof_end(e0, d, 0)
        ctx.dims = (N, H, W, C1, C2, Co)
        ctx.fslot, ctx.yslot, ctx.handles, ctx.split = fslot, yslot, handles, split
        ctx.set_materialize_grads(False)
        ctx.trainable = _sink_trainable(wv), _sink_trainable(bv)
        wb_low = None
        if need_bwd and f.requires_grad:
            d2 = _desc(N, H, W, C2, Co, 1, 1, 1)
            low = lowp if lowp is not None else wv[:, :, C1:, :]
            _, wb_low = packed_weights(d2, low, lowp, True)
        ctx.save_for_backward(a, f, out, wb_low)
        return out

    @staticmethod
    def backward(ctx, dy):
        a, f, out, wb_low = ctx.saved_tensors
        N, H, W, C1, C2, Co = ctx.dims
        dev = a.device
        M = N * H * W
        g = ctx.yslot.take() if ctx.yslot is not None else None          # slot deliveries are already multiplied by (out > 0)
        if dy is not None:
            dy = dy.contiguous().clone()
            call("danhip_relu_bwd_bias_grad", ptr(dy), ptr(out), None, M, Co, stream())
            g = dy if g is None else g.add_(dy)
        if g is None:
            return (None,) * 8
        wvp, bvp, lowp = ctx.handles
        c1, o1 = ctx.split
        need_dw = ctx.needs_input_grad[6] or ctx.trainable[0]
        need_db = ctx.needs_input_grad[7] or ctx.trainable[1]
        # ---- f's data gradient first (it heads the critical path): dF = g . W[C1:, :]^T, delivered into f's slot (+ its ReLU backward)
        df = None
        if ctx.needs_input_grad[1]:
            d2 = _desc(N, H, W, C2, Co, 1, 1, 1)
            ws, nws = _conv_scratch(d2, 1, dev)
            e0 = _prof_begin()
            xs = ctx.fslot
            if xs is not None:
                buf, acc = xs.target()
                call("danhip_conv2d_bwd_data_ws", ctypes.byref(d2), ptr(g), ptr(wb_low), ptr(f) if xs.is_relu else None, ptr(buf), acc, ptr(ws), nws, stream())
            else:
                df = torch.empty_like(f)
                call("danhip_conv2d_bwd_data_ws", ctypes.byref(d2), ptr(g), ptr(wb_low), None, ptr(df), 0, ptr(ws), nws, stream())
            _prof_end(e0, d2, 5 if (xs is not None and xs.is_relu) else 1)
        dwv = dbv = None
        if need_dw:
            sink = _grad_sink(wvp) if wvp is not None else None
            dw = sink if sink is not None else torch.zeros((1, 1, C1 + C2, Co), dtype=torch.float32, device=dev)
            db = None
            if need_db:
                bs = _grad_sink(bvp) if bvp is not None else None
                db = bs if bs is not None else torch.zeros(Co, dtype=torch.float32, device=dev)
                if bs is None:
                    dbv = db
            if sink is None:
                dwv = dw
            # row block of a (+ the bias gradient: column sums of g), row block of f; then the columns of the OTHER part go back to zero
            _wgrad_launch(_desc(N, H, W, C1, Co, 1, 1, 1), a, g, dw[:, :, :C1, :], db, C1, (g,))
            _wgrad_launch(_desc(N, H, W, C2, Co, 1, 1, 1), f, g, dw[:, :, C1:, :], None, C2, (g,))
            if c1 is not None:
                side = _CTX.wgrad["side"] if _CTX.wgrad["on"] else None
                with torch.cuda.stream(side) if side is not None else contextlib.nullcontext():
                    dw[:, :, :c1, o1:].zero_()
                    dw[:, :, c1:, :o1].zero_()
        elif need_db:
            raise NotImplementedError("a trainable bias without its kernel (no reference graph has one)")
        if _CTX.GRAD_READY_HOOK is not None and wvp is not None:
            _CTX.GRAD_READY_HOOK(wvp)
        return None, df, None, None, None, None, dwv, dbv


def concat_conv1x1_relu(a, f, wv, bv, split=None, trace_params=None):
    """relu(conv1x1(concat([a, f]), wv) + bv) without the concatenation; a is treated as a constant (stop_gradient).  wv: [1, 1, Ca + Cf, Co]
    (a FlatParams block-diagonal block, or any tensor); split = (rows, columns) of its upper-left block when wv is block-diagonal - the
    off-diagonal blocks of its gradient are then cleared.  trace_params: (w1, w2) kernel Parameters of the two parts (ops.TRACE: tests)."""
    assert not (_is_limbs(a) or _is_limbs(f)), "concat_conv1x1_relu is a 16-bit op (the nets take the unfused mix on the fp32 / split paths)"
    track = torch.is_grad_enabled() and (f.requires_grad or wv.requires_grad or _sink_trainable(wv))
    wvp = wv if (isinstance(wv, torch.nn.Parameter) or hasattr(wv, "_danhip_grad")) else None
    bvp = bv if (isinstance(bv, torch.nn.Parameter) or hasattr(bv, "_danhip_grad")) else None
    lowp = getattr(wv, "_danhip_lower", None)
    yslot = _new_slot(track)
    out = _ConcatMix.apply(a.detach(), f, _slot_of(f) if (track and f.requires_grad) else None, yslot, (wvp, bvp, lowp),
                           split if split is not None else (None, None), wv, bv)
    if _CTX.TRACE is not None and trace_params is not None and split is not None:
        _CTX.TRACE[id(trace_params[0])] = out.detach()[..., :split[1]].contiguous()
        _CTX.TRACE[id(trace_params[1])] = out.detach()[..., split[1]:].contiguous()
    if yslot is not None:
        yslot.__init__(out, True)
        out._dh_slot = yslot
    return out


class _BatchNorm(torch.autograd.Function):
    """tf.layers.batch_normalization(training=True) (+ optional ReLU) over NHWC bf16 — net/sfd_net.py:91-119."""

    @staticmethod
    def forward(ctx, x, gamma, beta, moving_mean, moving_var, eps, momentum, relu):
        C = x.shape[-1]
        M = x.numel() // C
        y = torch.empty_like(x)
        mean = torch.empty(C, dtype=torch.float32, device=x.device)
        rstd = torch.empty(C, dtype=torch.float32, device=x.device)
        ws = torch.empty(2 * C, dtype=torch.float32, device=x.device)
        call("danhip_batchnorm_fwd_train", ptr(x), ptr(gamma.detach()), ptr(beta.detach()), ptr(y), ptr(mean), ptr(rstd), ptr(moving_mean), ptr(moving_var),
             M, C, float(eps), float(momentum), int(relu), ptr(ws), stream())
        ctx.save_for_backward(x, gamma.detach(), mean, rstd, y if relu else None)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, dy):
        x, gamma, mean, rstd, y = ctx.saved_tensors
        C = x.shape[-1]
        M = x.numel() // C
        dy = dy.contiguous()
        if ctx.relu:
            dy = dy.clone()
            call("danhip_relu_bwd_bias_grad", ptr(dy), ptr(y), None, M, C, stream())
        dx = torch.empty_like(x)
        dgamma = torch.empty(C, dtype=torch.float32, device=x.device)
        dbeta = torch.empty(C, dtype=torch.float32, device=x.device)
        call("danhip_batchnorm_bwd", ptr(x), ptr(dy), ptr(gamma), ptr(mean), ptr(rstd), ptr(dx), ptr(dgamma), ptr(dbeta), M, C, stream())
        return dx, dgamma, dbeta, None, None, None, None, None


def batch_norm_train(x, gamma, beta, moving_mean=None, moving_var=None, eps=1e-5, momentum=0.997, relu=False):
    assert not _is_limbs(x), "batch_norm_train: a 16-bit training op"
    return _BatchNorm.apply(x, gamma, beta, moving_mean, moving_var, eps, momentum, relu)


def batch_norm_infer(x, gamma, beta, moving_mean, moving_var, eps=1e-5, relu=False):
    x = _f32_in(x)
    C = x.shape[-1]
    y = torch.empty_like(x)
    rstd = torch.rsqrt(moving_var + eps)
    call("danhip_batchnorm_fwd_infer", ptr(x), ptr(gamma.detach()), ptr(beta.detach()), ptr(moving_mean), ptr(rstd), ptr(y), x.numel() // C, C, int(relu), stream())
    return y


class _DeformSample(torch.autograd.Function):
    """Deformable im2col (cpp/Deform/deform_conv.cu:229-275) and its backward (col2im :281-328, col2im_coord :335-389):
    x bf16 [N,H,W,C], offsets bf16 [N,Ho,Wo,dg*2*kh*kw] -> S bf16 [N,Ho,Wo,kh*kw*C] (k = tap*C + c)."""

    @staticmethod
    def forward(ctx, x, offsets, kh, kw, stride, dilation, dg):
        N, H, W, C = x.shape
        Ho, Wo = -(-H // stride), -(-W // stride)
        assert x.dtype == ACT and offsets.dtype == ACT and x.is_contiguous() and offsets.is_contiguous()
        assert offsets.shape == (N, Ho, Wo, dg * 2 * kh * kw), (offsets.shape, (N, Ho, Wo, dg * 2 * kh * kw))
        S = torch.empty((N, Ho, Wo, kh * kw * C), dtype=ACT, device=x.device)
        call("danhip_deform_sample_fwd", ptr(x), ptr(offsets), ptr(S), N, H, W, C, kh, kw, stride, dilation, dg, stream())
        ctx.save_for_backward(x, offsets)
        ctx.cfg = (kh, kw, stride, dilation, dg)
        return S

    @staticmethod
    def backward(ctx, dS):
        x, offsets = ctx.saved_tensors
        kh, kw, stride, dilation, dg = ctx.cfg
        N, H, W, C = x.shape
        dx = torch.empty_like(x)
        doff = torch.empty_like(offsets)
        ws = torch.empty(x.numel() + 64, dtype=torch.float32, device=x.device)      # scatter target + the far-corner statistic
        call("danhip_deform_sample_bwd", ptr(x), ptr(offsets), ptr(dS.contiguous()), ptr(dx), ptr(doff), N, H, W, C, kh, kw, stride, dilation, dg, 0,
             ptr(ws), ws.numel() * 4, stream())
        return dx, doff, None, None, None, None, None


# (KEEP_DEFORM_COL: a field of OpsContext, see the top of the module)


class _DeformConv(torch.autograd.Function):
    """DeformConvOp / DeformConvBackpropOp (cpp/Deform/deform_conv.cc:392-535, :635-771) through the single-call entry
    points.  KEEP_DEFORM_COL (default on): the forward's im2col buffer (9x the activation, 1.9 GB at 160x160x256 batch 16) is kept for
    the weight gradient instead of being re-sampled in backward as the reference does on its 11 GB cards (deform_conv.cc:744-748).
    x bf16 [N,H,W,C]; w fp32 [1,1,kh*kw*C,Cout] (the OIHW variable viewed as the GEMM operand); offsets bf16."""

    @staticmethod
    def forward(ctx, x, w, b, offsets, kh, kw, stride, dilation, dg, relu, b_param, yslot, xslot=None, w_param=None):
        N, H, W, C = x.shape
        cout = w.shape[-1]
        ctx.xslot = xslot
        ctx.w_param, ctx.block_w = w_param, _sink_trainable(w)
        assert x.dtype == ACT and offsets.dtype == ACT and x.is_contiguous() and offsets.is_contiguous()
        assert w.shape == (1, 1, kh * kw * C, cout) and cout % 8 == 0
        Ho, Wo = -(-H // stride), -(-W // stride)
        assert offsets.shape == (N, Ho, Wo, dg * 2 * kh * kw), (offsets.shape, (N, Ho, Wo, dg * 2 * kh * kw))
        d = _desc(N, Ho, Wo, kh * kw * C, cout, 1, 1, 1)
        need_bwd = w.requires_grad or x.requires_grad or offsets.requires_grad or ctx.block_w
        # (w_param: the GEMM operand is a block of the trainer's flat buffers or a cached gradient-free copy: packed once per weight version)
        wf, wb = packed_weights(d, w, w_param, need_bwd) if w_param is not None else pack_conv_weight(d, w.detach().contiguous(), need_bwd=need_bwd)
        y = torch.empty((N, Ho, Wo, cout), dtype=ACT, device=x.device)
        keep = _CTX.KEEP_DEFORM_COL and need_bwd
        if not keep and _lib.lib().danhip_deform_conv_fused(N, H, W, C, cout, kh, kw, stride, dg):
            ws, nws = None, 0                            # the fused kernel samples straight into the GEMM's LDS tile: no column buffer
        else:
            nws = _lib.lib().danhip_deform_conv_workspace_bytes(N, H, W, C, kh, kw, stride, 0)
            ws = torch.empty(nws, dtype=torch.uint8, device=x.device)
        call("danhip_deform_conv_fwd", ptr(x), ptr(wf), ptr(b.detach()) if b is not None else None, ptr(offsets), ptr(y), N, H, W, C, cout, kh, kw,
             stride, dilation, dg, int(relu), ptr(ws) if ws is not None else None, nws, stream())
        ctx.cfg = (kh, kw, stride, dilation, dg, cout, relu)
        ctx.b_param, ctx.yslot, ctx.has_bias = b_param, yslot, b is not None
        ctx.set_materialize_grads(False)
        ctx.save_for_backward(x, offsets, wb, y if relu else None)
        ctx.col = ws if keep else None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, offsets, wb, y = ctx.saved_tensors
        kh, kw, stride, dilation, dg, cout, relu = ctx.cfg
        N, H, W, C = x.shape
        col, ctx.col = ctx.col, None
        g = ctx.yslot.take() if ctx.yslot is not None else None          # slot deliveries arrive ReLU-masked
        if dy is not None:
            assert dy.dtype == ACT and dy.shape[-1] == cout
            dy = dy.contiguous()
            if relu:
                dy = dy.clone()
                call("danhip_relu_bwd_bias_grad", ptr(dy), ptr(y), None, dy.numel() // cout, cout, stream())
            g = dy if g is None else g.add_(dy)
        if g is None:
            return (None,) * 14
        bp = ctx.b_param
        db_sink = _grad_sink(bp) if bp is not None else None
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = db_sink if db_sink is not None else torch.zeros(cout, dtype=torch.float32, device=x.device)
        # x's gradient goes straight into its producer's slot - times (x > 0) when x is a ReLU output, added to what other consumers (the
        # offset convolution) delivered - instead of through an autograd tensor the producer would clone, mask and add (three map-sized passes)
        xs = ctx.xslot if ctx.needs_input_grad[0] else None
        if xs is not None:
            dx, acc = xs.target()
            relu_x = 1 if xs.is_relu else 0
        else:
            dx, acc, relu_x = torch.empty_like(x), 0, 0
        doff = torch.empty_like(offsets)
        dw_sink = _grad_sink(ctx.w_param) if ctx.w_param is not None else None       # the flat gradient buffer's block: accumulated in place
        dw = dw_sink if dw_sink is not None else torch.zeros((1, 1, kh * kw * C, cout), dtype=torch.float32, device=x.device)
        nws = _lib.lib().danhip_deform_conv_workspace_bytes(N, H, W, C, kh, kw, stride, 1)
        ws = torch.empty(nws, dtype=torch.uint8, device=x.device)
        call("danhip_deform_conv_bwd_deliver", ptr(x), ptr(wb), ptr(offsets), ptr(g), ptr(col), ptr(dx), ptr(doff), ptr(dw), ptr(db), N, H, W, C, cout,
             kh, kw, stride, dilation, dg, acc, relu_x, ptr(ws), nws, stream())
        if _CTX.GRAD_READY_HOOK is not None and bp is not None:
            _CTX.GRAD_READY_HOOK(bp)
        if _CTX.GRAD_READY_HOOK is not None and dw_sink is not None:
            _CTX.GRAD_READY_HOOK(ctx.w_param)
        return ((None if xs is not None else dx), (None if dw_sink is not None else dw), (None if db_sink is not None else db), doff, None, None, None, None,
                None, None, None, None, None, None)


def deform_conv(x, w1x1, b, offsets, kh, kw, stride=1, dilation=1, deformable_group=1, relu=False):
    """y = act(DeformConvOp(x, filter, offsets) + b); w1x1 = the filter viewed [1,1,kh*kw*C,Cout]."""
    x, offsets = _f32_in(x, offsets)
    if _f32_infer(x):
        N, H, W, C = x.shape
        Ho, Wo = -(-H // stride), -(-W // stride)
        S = torch.empty((N, Ho, Wo, kh * kw * C), dtype=torch.float32, device=x.device)
        call("danhip_deform_sample_fwd_f32", ptr(x.contiguous()), ptr(offsets.contiguous()), ptr(S), N, H, W, C, kh, kw, stride, dilation, deformable_group,
             stream())
        return (_conv2d_split if _CTX.SPLIT_EVAL else _conv2d_f32)(S, w1x1, b, 1, relu, None, "same")
    bp = b if isinstance(b, torch.nn.Parameter) else None
    wp = w1x1 if hasattr(w1x1, "_danhip_grad") else None
    track = torch.is_grad_enabled() and (x.requires_grad or w1x1.requires_grad or offsets.requires_grad or _sink_trainable(w1x1))
    yslot = _new_slot(track)
    y = _DeformConv.apply(x, w1x1, b, offsets, kh, kw, stride, dilation, deformable_group, relu, bp, yslot,
                          _slot_of(x) if (track and x.requires_grad) else None, wp)
    if yslot is not None:
        yslot.__init__(y, relu)
        y._dh_slot = yslot
    return y


def preprocess_f32(img_rgb_u8):
    """uint8 RGB [N,H,W,3] -> fp32 [N,H,W,3] BGR - mean (dan_preprocessing.py:55-57,755-758): exact fp32 subtraction, 3 real channels."""
    assert img_rgb_u8.dtype == torch.uint8 and img_rgb_u8.shape[-1] == 3
    means = torch.tensor([123.68, 116.78, 103.94], dtype=torch.float32, device=img_rgb_u8.device)
    return (img_rgb_u8.to(torch.float32) - means).flip(-1).contiguous()


def deform_sample(x, offsets, kh, kw, stride=1, dilation=1, deformable_group=1):
    x, offsets = _f32_in(x, offsets)
    return _DeformSample.apply(x, offsets, kh, kw, stride, dilation, deformable_group)


def face_scores(cls, threshold=None):
    """softmax(cls)[..., 1] of two-way fp32 logits [..., 2] (eval_dan.py:356, eval_sfd.py:281) -> scores [...]; with `threshold` also the int32
    mask (score > threshold) (train_dan.py:438-439): -> (scores, mask).  One libdanhip launch (no torch.softmax on the product path)."""
    assert cls.dtype == torch.float32 and cls.shape[-1] == 2
    c = cls.contiguous()
    n = c.numel() // 2
    score = torch.empty(c.shape[:-1], dtype=torch.float32, device=c.device)
    mask = torch.empty(c.shape[:-1], dtype=torch.int32, device=c.device) if threshold is not None else None
    call("danhip_face_scores", ptr(c), ptr(score), ptr(mask), float(threshold or 0.0), n, stream())
    return (score, mask) if threshold is not None else score


def argsort_desc(scores, ties_high_index_first=False):
    """Positions of a 1-D fp32 vector by descending value, equal values by ascending index — torch.sort(descending=True, stable=True).indices,
    i.e. tf.nn.top_k's / tf.image.non_max_suppression's candidate order (utility/bbox_util.py:61-91); ties_high_index_first: numpy's
    argsort()[::-1] (eval_dan.py:255).  -> int64 indices.  libdanhip's bitonic arg-sort (csrc/sort.hip): no torch.sort on the product path."""
    assert scores.dim() == 1 and scores.dtype == torch.float32
    n = scores.shape[0]
    idx = torch.empty((n,), dtype=torch.int32, device=scores.device)
    if n == 0:
        return idx.long()
    s = scores.contiguous()
    nbytes = int(_lib.lib().danhip_argsort_workspace_bytes(n))
    ws = torch.empty((nbytes,), dtype=torch.uint8, device=scores.device)
    call("danhip_argsort_desc_f32", ptr(s), n, 1 if ties_high_index_first else 0, ptr(idx), ptr(ws), nbytes, stream())
    return idx.long()


# ---- `ops.USE_SLOTS`, `ops.TRACE = {}` ... (rounds 1-4 spelling; tests, bench.py, tools): reads and writes go to the active context
class _OpsModule(type(os)):
    def __getattr__(self, name):
        if name in _CTX_FIELDS:
            return getattr(_CTX, name)
        raise AttributeError("module %r has no attribute %r" % (self.__name__, name))

    def __setattr__(self, name, value):
        if name in _CTX_FIELDS:
            setattr(_CTX, name, value)
        else:
            super().__setattr__(name, value)


import sys as _sys  # noqa: E402
_sys.modules[__name__].__class__ = _OpsModule
