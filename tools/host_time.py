"""Host issue time against GPU time of one eager training step: is the step launch-bound?  usage: python tools/host_time.py [sfd|pb|dan|dan_deform] [profile|aten|layers]
"aten": every torch-native operator that put work on the GPU in one step (copies, fills, adds, cats ...), by Python call site - the launches
that are not libdanhip kernels.  "layers" (run with DANHIP_WGRAD_STREAM=0): event-bracketed time of every convolution launch of one step by
(pass, shape) - which layers the small-kernel time belongs to."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dan_amd import synthetic

which = sys.argv[1] if len(sys.argv) > 1 else "sfd"
dev = torch.device("cuda:0")
B, S = 16, 640
imgs = synthetic.make_images(B, S, S, dev, seed=1)
gts = synthetic.make_gt_boxes(B, S, S, seed=2)
if which == "sfd":
    from dan_amd.train_sfd import AnchorConfig, SFDModel, SFDTrainer
    anchors = AnchorConfig(S, S, dev)
    loc_t, cls_t, _ = anchors.encode_batch(gts)
    tr = SFDTrainer(SFDModel(device=dev), world=1)
    args = (imgs, loc_t, cls_t)
elif which == "pb":
    from dan_amd.train_pb import PBAnchorTargets, PBModel, PBTrainer
    tr = PBTrainer(PBModel(device=dev), world=1)
    args = (imgs, PBAnchorTargets(S, S, dev).encode_batch(gts))
else:
    from dan_amd.train_dan import DANModel, DANTrainer, dan_anchor_config, encode_batch_dan
    anchors = dan_anchor_config(S, S, dev)
    tr = DANTrainer(DANModel(device=dev, deform=which == "dan_deform"), anchors, world=1)
    args = (imgs,) + encode_batch_dan(anchors, gts)
for _ in range(3):
    tr.train_step(*args)
torch.cuda.synchronize()
for rep in range(4):
    t0 = time.perf_counter()
    tr.train_step(*args)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("%s: host issue %.2f ms, until GPU done %.2f ms" % (which, (t1 - t0) * 1e3, (t2 - t0) * 1e3))
if len(sys.argv) > 2 and sys.argv[2] == "aten":
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True, record_shapes=True) as prof:
        tr.train_step(*args)
        torch.cuda.synchronize()
    for e in prof.key_averages(group_by_input_shape=True):       # which tensors the engine's own adds / copies work on
        if e.key in ("aten::add_", "aten::add", "aten::copy_", "aten::clone") and e.self_device_time_total > 0:
            print("%4d x %-12s %8.1f us GPU   shapes %s" % (e.count, e.key, e.self_device_time_total, e.input_shapes))
    rows = [e for e in prof.key_averages(group_by_stack_n=12) if e.key.startswith("aten::") and e.self_device_time_total > 0]
    rows.sort(key=lambda e: -e.count)
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    total = 0
    for e in rows:
        site = [f for f in e.stack if here in f and "host_time.py" not in f]
        total += e.count
        print("%4d x %-28s %8.1f us GPU   %s" % (e.count, e.key, e.self_device_time_total, " <- ".join(x.replace(here + "/", "") for x in site[:3])))
    print("torch-native GPU launches per step:", total)
    # single weight packs (a conv whose kernel is not a cached Parameter packs on the spot): by call site
    import collections
    import traceback
    from dan_amd import ops
    sites = collections.Counter()
    real_call = ops.call

    def counting_call(name, *a):
        if name == "danhip_pack_conv_weight":
            fr = [f for f in traceback.extract_stack()[:-1] if here in f.filename and "host_time.py" not in f.filename]
            sites[" <- ".join("%s:%d" % (f.filename.replace(here + "/", ""), f.lineno) for f in fr[-4:][::-1])] += 1
        return real_call(name, *a)

    ops.call = counting_call
    # ... and who calls the torch-native in-place adds / clones / fills / copies of 16-bit activations (the profiler gives no Python stacks here)
    tsites = collections.Counter()
    patched = {}

    def wrap(name):
        real = getattr(torch.Tensor, name)

        def f(self, *a, **k):
            if self.is_cuda and self.numel() >= 1 << 16:
                fr = [x for x in traceback.extract_stack()[:-1] if here in x.filename and "host_time.py" not in x.filename]
                tsites[(name, tuple(self.shape), str(self.dtype).replace("torch.", ""),
                        " <- ".join("%s:%d" % (x.filename.replace(here + "/", ""), x.lineno) for x in fr[-3:][::-1]))] += 1
            return real(self, *a, **k)

        patched[name] = real
        setattr(torch.Tensor, name, f)

    for nm in ("add_", "clone", "zero_", "copy_", "contiguous"):
        wrap(nm)
    tr.train_step(*args)
    torch.cuda.synchronize()
    for nm, real in patched.items():
        setattr(torch.Tensor, nm, real)
    ops.call = real_call
    for k, v in sites.most_common():
        print("%4d x danhip_pack_conv_weight   %s" % (v, k))
    for (name, shp, dt, site), v in sorted(tsites.items(), key=lambda kv: -kv[1]):
        print("%4d x Tensor.%-10s %-22s %-9s %s" % (v, name, shp, dt, site))
elif len(sys.argv) > 2 and sys.argv[2] == "layers":
    import collections
    from dan_amd import ops
    rec = []
    real_end = ops._prof_end

    def end(e0, d, which, st=None):
        if e0 is None:
            return
        e1 = torch.cuda.Event(enable_timing=True)
        e1.record(st if st is not None else torch.cuda.current_stream())
        rec.append((e0, e1, {0: "fwd", 4: "fwd+pool", 1: "dgrad", 5: "dgrad+mask", 2: "wgrad"}[which], (d.H, d.W, d.Cin, d.Cout, d.kh, d.kw, d.stride)))

    ops._prof_end = end
    ops.PROFILE = {}
    tr.train_step(*args)
    torch.cuda.synchronize()
    ops.PROFILE = None
    ops._prof_end = real_end
    by = collections.defaultdict(lambda: [0.0, 0])
    for e0, e1, kind, shp in rec:
        by[(kind, shp)][0] += e0.elapsed_time(e1)
        by[(kind, shp)][1] += 1
    tot = sum(v[0] for v in by.values())
    print("convolution launches: %d, %.2f ms" % (len(rec), tot))
    for (kind, shp), (ms, n) in sorted(by.items(), key=lambda kv: -kv[1][0])[:70]:
        print("%8.3f ms %3d x  %-10s H %4d W %4d  %4d -> %4d  %dx%d /%d" % ((ms, n, kind) + shp))
    grp = collections.defaultdict(lambda: [0.0, 0])
    for (kind, shp), (ms, n) in by.items():
        key = "%dx%d %d->%d" % (shp[4], shp[5], shp[2], shp[3])
        grp[(kind.split("+")[0], key)][0] += ms
        grp[(kind.split("+")[0], key)][1] += n
    print("by (pass, kernel size, channels) over all levels:")
    for (kind, key), (ms, n) in sorted(grp.items(), key=lambda kv: -kv[1][0])[:60]:
        print("%8.3f ms %3d x  %-6s %s" % (ms, n, kind, key))
elif len(sys.argv) > 2:
    import cProfile
    import pstats
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(5):
        tr.train_step(*args)
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(25)
