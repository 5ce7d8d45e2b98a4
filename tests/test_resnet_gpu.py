"""ResNet-50 + batch-norm backbone (net/resnet_danet.py:92-228) on the HIP path against the oracle graph with
identical variables: 'valid' 7x7/2 stem on the explicitly padded image, 3x3/2 'same' max-pool, bottleneck stages with projection
shortcuts, batch norm with moving statistics (inference) and with batch statistics (training), and a gradient smoke test."""
import pytest
import torch

from oracle import nets as ON
from oracle import tf_ops as T

pytestmark = pytest.mark.gpu


def _setup(dev, training, size=64):
    from dan_amd import synthetic
    from dan_amd.net import resnet_danet, sfd_net
    from dan_amd.net.variables import VariableStore
    imgs = synthetic.make_images(2, size, size, "cpu", seed=17)
    x = ON.preprocess_synthetic(imgs)
    P = ON.Params(create=True, seed=77)
    with torch.no_grad():
        ON.resnet_get_featmaps(P, x, 50, training)
    g = torch.Generator().manual_seed(3)
    for n in P.t:                                   # non-trivial BN parameters / statistics
        if n.endswith("/bn/gamma"):
            P.t[n] = 0.5 + torch.rand(P.t[n].shape, generator=g)
        elif n.endswith("/bn/beta") or n.endswith("/bn/moving_mean"):
            P.t[n] = 0.2 * torch.randn(P.t[n].shape, generator=g)
        elif n.endswith("/bn/moving_variance"):
            P.t[n] = 0.5 + torch.rand(P.t[n].shape, generator=g)
    vs = VariableStore(device=dev)
    net = resnet_danet.ResNetBackbone(50, variables=vs)
    xin = sfd_net.prepare_input(imgs.to(dev))
    with torch.no_grad():
        net.get_featmaps(xin, training=False)       # creates the variables and buffers
    vs.load_tf_named({n: t for n, t in P.t.items() if "/moving_" not in n})
    for n, t in P.t.items():
        if "/moving_" in n:
            vs.bufs[n].copy_(t.to(dev))
    assert set(n for n, _ in vs.named()) | set(vs.bufs) == set(P.t.keys())
    return P, net, xin, x


@pytest.mark.parametrize("training", [False, True])
def test_resnet50_featmaps_parity(training, dev):
    # batch statistics over a handful of positions amplify bf16 rounding (the variance of 2 x 2 x 2 samples is itself noisy): the
    # training-mode comparison runs at 128 x 128 and stops at the stride-16 map; inference mode (moving statistics) checks all six
    size = 128 if training else 64
    P, net, xin, x = _setup(dev, training, size)
    with torch.no_grad():
        ref = ON.resnet_get_featmaps(ON.Params(P.t, emulate_bf16=True), x.to(torch.bfloat16).float(), 50, training)
        got = net.get_featmaps(xin, training=training)
    q = size // 4
    assert [tuple(f.shape[1:]) for f in got] == [(q, q, 256), (q // 2, q // 2, 512), (q // 4, q // 4, 1024), (q // 8, q // 8, 2048),
                                                (q // 16, q // 16, 512), (max(q // 32, 1), max(q // 32, 1), 256)]
    for i, (a, r) in enumerate(zip(got, ref)):
        assert a.shape == r.shape and torch.isfinite(a.float()).all()
        if training and i >= 3:
            continue
        scale = r.abs().max().item()
        err = (a.float().cpu() - r).abs().max().item()
        assert err <= (0.08 if not training else 0.15) * scale + 1e-2, (i, err, scale)


def test_stem_ops_against_oracle(dev):
    """The two new pieces alone: 'valid' 7x7 stride-2 convolution (fwd, dx, dw) and the 3x3/2 'same' max-pool (fwd, first-max backward)."""
    from dan_amd import ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn((2, 29, 35, 8), generator=g).to(torch.bfloat16)
    w = (torch.randn((7, 7, 8, 64), generator=g) / 20).to(torch.bfloat16).float()
    xr, wr = x.float().requires_grad_(True), w.clone().requires_grad_(True)
    ref = T.conv2d_valid(xr, wr, None, stride=2)
    dy = torch.randn(ref.shape, generator=g).to(torch.bfloat16)
    ref.backward(dy.float())
    xd, wd = x.to(dev).requires_grad_(True), w.to(dev).requires_grad_(True)
    y = ops.conv2d(xd, wd, None, stride=2, padding="valid")
    y.backward(dy.to(dev))
    assert y.shape == ref.shape == (2, 12, 15, 64)
    for got, want in ((y, ref), (xd.grad, xr.grad), (wd.grad, wr.grad)):
        want = want.detach()
        assert (got.detach().float().cpu() - want).abs().max().item() <= 2.0 ** -6 * want.abs().max().item() + 2e-3
    for h, wdt in ((16, 16), (15, 17), (7, 8), (1, 5)):
        p = torch.randn((2, h, wdt, 16), generator=g).to(torch.bfloat16)
        pr = p.float().requires_grad_(True)
        ref = T.max_pool_3x3_s2_same(pr)
        dyp = torch.randn(ref.shape, generator=g).to(torch.bfloat16)
        ref.backward(dyp.float())
        pd = p.to(dev).requires_grad_(True)
        out = ops.max_pool_3x3_s2(pd)
        out.backward(dyp.to(dev))
        assert torch.equal(out.float().cpu(), ref.detach())
        assert (pd.grad.float().cpu() - pr.grad).abs().max().item() <= 2.0 ** -7 * pr.grad.abs().max().item() + 1e-6   # sums of up to 4 bf16 terms


def test_resnet50_backward_smoke(dev):
    P, net, xin, x = _setup(dev, True)
    feats = net.get_featmaps(xin, training=True)
    loss = sum(f.float().pow(2).mean() for f in feats)
    loss.backward()
    grads = [p.grad for _, p in net.vs.named() if p.grad is not None]
    assert len(grads) > 150 and all(torch.isfinite(g).all() for g in grads)
    assert sum(float(g.abs().sum()) for g in grads) > 0
