"""Block-level parity of DAN's fused context module and stage-2 input mix against the CPU oracle: the fused forms
(ops._ContextBlock: ONE autograd node over channel-slice views, the pool branch computed as avg(conv(x)); ops._ConcatMix: one 1x1 with a
block-diagonal kernel over the never-written concatenation) used to rest on HIP-vs-HIP comparisons (tests/test_context_block_gpu.py,
tests/test_concat_mix_gpu.py) plus the whole-graph tests.  Here each block alone is compared with oracle/nets.py
(se_inception_block_v1 <- net/danet.py:842-918, get_features_stage2 <- net/danet.py:931-954) in 16-bit-storage emulation: forward,
input gradient and every variable's gradient, with the HIP forward's ReLU decisions imposed on the oracle (a pre-activation within
rounding of zero must not decide the comparison).  Bounds: forward 2^-6 of the output scale, gradients 0.05 relative L2."""
import pytest
import torch

from oracle import nets as ON

pytestmark = pytest.mark.gpu


def _rel(a, b):
    return (a.reshape(-1) - b.reshape(-1)).norm().item() / (b.norm().item() + 1e-12)


def _make_params(fwd, seed):
    P = ON.Params(create=True, seed=seed)
    with torch.no_grad():
        fwd(P)
    g = torch.Generator().manual_seed(seed + 1)
    for n in P.t:                                           # non-zero biases (the reference initialises them to zero)
        if n.endswith("/bias"):
            P.t[n] = 0.1 * torch.randn(P.t[n].shape, generator=g)
    return P


def _hip_run(fn, vs, inputs, dy, dev):
    """-> (out, [input grads], {var: grad}, impose dict) of the HIP path with its ReLU decisions recorded."""
    from dan_amd import ops
    from gradcheck import collect_trace
    named = vs.named()
    for _, p in named:
        p.grad = None
    xs = [t.clone().requires_grad_(True) for t in inputs]
    ops.TRACE = {}
    try:
        out = fn(*xs)
        rec, ops.TRACE = ops.TRACE, None
    finally:
        ops.TRACE = None
    trace = collect_trace(named, rec, {})
    out.backward(dy)
    torch.cuda.synchronize()
    return (out.detach().float().cpu(), [None if t.grad is None else t.grad.float().cpu() for t in xs],
            {n: (None if p.grad is None else p.grad.detach().float().cpu()) for n, p in named}, trace)


def _oracle_run(fn, P, inputs, dy, impose):
    params = {n: v.clone().requires_grad_(True) for n, v in P.t.items()}
    PO = ON.Params(params, emulate_bf16=True)
    PO.impose = impose
    xs = [t.clone().requires_grad_(True) for t in inputs]
    out = fn(PO, *xs)
    (out * dy).sum().backward()
    return out.detach(), [t.grad for t in xs], {n: p.grad for n, p in params.items()}


@pytest.mark.parametrize("N,H,W,C", [(2, 40, 40, 256), (1, 37, 45, 256), (2, 12, 20, 512), (2, 5, 5, 1024)])
def test_fused_context_block_against_the_oracle_block(N, H, W, C, dev):
    from dan_amd import ops
    from dan_amd.net import danet
    from dan_amd.net.variables import VariableStore
    g = torch.Generator().manual_seed(N * 100 + H)
    x = torch.randn((N, H, W, C), generator=g).to(ops.ACT)
    dy = torch.randn((N, H, W, C), generator=g).to(ops.ACT)
    P = _make_params(lambda P_: ON.se_inception_block_v1(P_, x.float(), "blk"), 21)
    vs = VariableStore(device=dev, seed=1)
    vs.load_tf_named(P.t)
    bb = danet.VGG16Backbone("channels_last", variables=vs)
    assert bb.FUSED_CONTEXT_BLOCK
    out, (dx,), grads, trace = _hip_run(lambda t: bb.se_inception_block(t, "blk"), vs, [x.to(dev)], dy.to(dev), dev)
    assert len(trace["relu"]) == 10, sorted(trace["relu"])          # the block's ten ReLU layers were recorded under their kernel variables
    want, (dxo,), gwant = _oracle_run(lambda P_, t: ON.se_inception_block_v1(P_, t, "blk"), P, [x.float()], dy.float(), trace)
    assert (out - want).abs().max().item() <= 2.0 ** -6 * want.abs().max().item()
    assert _rel(dx, dxo) <= 0.05, _rel(dx, dxo)
    assert set(gwant) == set(grads) and len(gwant) == 20
    bad = [(n, round(_rel(grads[n], gwant[n]), 4)) for n in gwant if _rel(grads[n], gwant[n]) > 0.05]
    assert not bad, bad


@pytest.mark.parametrize("N,H,W,C", [(2, 40, 40, 256), (1, 21, 33, 256), (2, 10, 10, 512)])
def test_fused_stage2_mix_and_block_against_the_oracle(N, H, W, C, dev):
    """get_features_stage2 (net/danet.py:931-954) for one level: stop_gradient(stage-1 feature) -> 1x1 (C // 3, ReLU), backbone feature ->
    1x1 (C - C // 3, ReLU), concat, context block.  The stage-1 feature receives NO gradient; the 85- / 171-column (170 / 342) kernels are
    the diagonal blocks of one kernel on the HIP path."""
    from dan_amd import ops
    from dan_amd.net import danet
    from dan_amd.net.variables import VariableStore
    g = torch.Generator().manual_seed(N * 10 + W)
    s1 = torch.randn((N, H, W, C), generator=g).to(ops.ACT)
    f = torch.randn((N, H, W, C), generator=g).to(ops.ACT)
    dy = torch.randn((N, H, W, C), generator=g).to(ops.ACT)
    ofn = lambda P_, a, b: ON.get_features_stage2(P_, [a], [b], ON.se_inception_block_v1)[0]
    P = _make_params(lambda P_: ofn(P_, s1.float(), f.float()), 33)
    vs = VariableStore(device=dev, seed=1)
    vs.load_tf_named(P.t)
    bb = danet.VGG16Backbone("channels_last", variables=vs)
    assert bb.FUSED_STAGE2_MIX
    out, (ds1, df), grads, trace = _hip_run(lambda a, b: bb.get_features_stage2([a], [b])[0], vs, [s1.to(dev), f.to(dev)], dy.to(dev), dev)
    assert len(trace["relu"]) == 12                                  # the two mixing convolutions + the block's ten
    want, (ds1o, dfo), gwant = _oracle_run(ofn, P, [s1.float(), f.float()], dy.float(), trace)
    assert (out - want).abs().max().item() <= 2.0 ** -6 * want.abs().max().item()
    assert ds1o is None and (ds1 is None or ds1.abs().max().item() == 0.0)       # stop_gradient
    assert _rel(df, dfo) <= 0.05, _rel(df, dfo)
    assert set(gwant) == set(grads) and len(gwant) == 24
    bad = [(n, round(_rel(grads[n], gwant[n]), 4)) for n in gwant if _rel(grads[n], gwant[n]) > 0.05]
    assert not bad, bad
