"""Diagnosis for tests/test_grad_parity_gpu.py: per-variable relative gradient error of the HIP path against the bf16-emulating CPU oracle
for (a) the real loss and (b) a fixed random upstream gradient on the head outputs (no mining: isolates the backward graph)."""
import sys

import torch

sys.path.insert(0, ".")
from oracle import nets as ON, train as OT
from dan_amd import synthetic

which = sys.argv[1] if len(sys.argv) > 1 else "pb"
H, W, B = (int(sys.argv[2]), int(sys.argv[3]), 2) if len(sys.argv) > 3 else (64, 64, 2)
dev = torch.device("cuda:0")
imgs = synthetic.make_images(B, H, W, "cpu", seed=3)
x = ON.preprocess_synthetic(imgs)
deform = which == "dan_deform"
fwd = ON.pb_forward if which == "pb" else (lambda P, xx: ON.dan_forward(P, xx, deform=deform))
P = ON.Params(create=True, seed=11)
with torch.no_grad():
    fwd(P, x)
g = torch.Generator().manual_seed(99)
for n in P.t:
    if n.endswith("/bias"):
        P.t[n] = 0.05 * torch.randn(P.t[n].shape, generator=g)
    if deform and n.endswith("deform_conv/conv2d/kernel"):
        P.t[n] = 0.02 * torch.randn(P.t[n].shape, generator=g)
    if deform and n.endswith("deform_conv/conv2d/bias"):
        P.t[n] = 0.6 * torch.randn(P.t[n].shape, generator=g)
if which == "pb":
    from dan_amd.train_pb import PBModel, PBTrainer
    model = PBModel(device=dev)
    model.vs.load_tf_named(P.t)
    tr = PBTrainer(model, world=1)
    flat_out = lambda o: [o[k][j] for k in ("face", "head", "body") for j in (0, 1)]
else:
    from dan_amd.train_dan import DANModel, DANTrainer, dan_anchor_config
    model = DANModel(device=dev, deform=deform)
    model.vs.load_tf_named(P.t)
    tr = DANTrainer(model, dan_anchor_config(H, W, dev), world=1)
    flat_out = lambda o: [o[0][0], o[0][1], o[1][0], o[1][1]]
params = {n: v.clone().requires_grad_(True) for n, v in P.t.items()}
outs_ref = flat_out(fwd(ON.Params(params, emulate_bf16=True), x.to(torch.bfloat16).float()))
gen = torch.Generator().manual_seed(5)
Gs = [torch.randn(o.shape, generator=gen) for o in outs_ref]
sum((o * G).sum() for o, G in zip(outs_ref, Gs)).backward()
tr.flat.zero_grad()
outs = flat_out(model.forward(imgs.to(dev)))
for o, r in zip(outs, outs_ref):
    print("fwd rel err %.4f" % ((o.detach().cpu() - r.detach()).abs().max().item() / r.abs().max().item()))
torch.autograd.backward(outs, [G.to(dev) for G in Gs])
torch.cuda.synchronize()
rows = []
for name, prm in model.vs.named():
    want = params[name].grad
    got = prm.grad.detach().reshape(-1).cpu()
    if want is None:
        rows.append((name, -1.0, got.abs().max().item(), 0.0))
        continue
    want = want.reshape(-1)
    rows.append((name, (got - want).norm().item() / (want.norm().item() + 1e-12), got.norm().item(), want.norm().item()))
for r in rows:
    if r[1] > 0.05 or r[1] < 0:
        print("%-70s rel %.4f  |got| %.4g |want| %.4g" % r)
print("max rel", max(r[1] for r in rows), "n", len(rows))
