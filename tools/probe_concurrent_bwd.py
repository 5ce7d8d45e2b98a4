import os, sys, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dan_amd import ops
from dan_amd._lib import call, ptr
dev = torch.device("cuda:0")
def mk(N,H,W,Cin,Cout,k):
    g = torch.Generator().manual_seed(1)
    x = torch.randn((N,H,W,Cin), generator=g).to(torch.bfloat16).to(dev)
    w = (torch.randn((k,k,Cin,Cout), generator=g)/(k*k*Cin)**0.5).to(dev)
    d = ops._desc(N,H,W,Cin,Cout,k,k,1)
    wf, wb = ops.pack_conv_weight(d, w, need_bwd=True)
    dy = torch.randn((N,d.Ho,d.Wo,Cout), generator=g).to(torch.bfloat16).to(dev)
    dx = torch.empty_like(x); dw = torch.zeros((k,k,Cin,Cout), device=dev); db = torch.zeros(Cout, device=dev)
    return d,x,wb,dy,dx,dw,db
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
for name, shp in (("conv5_1",(16,40,40,512,512,3)),("fc6",(16,20,20,512,1024,3)),("fc7",(16,20,20,1024,1024,1)),("conv4_2",(16,80,80,512,512,3))):
    d,x,wb,dy,dx,dw,db = mk(*shp)
    def dgrad(st): call("danhip_conv2d_bwd_data", ctypes.byref(d), ptr(dy), ptr(wb), ptr(x), ptr(dx), 0, ctypes.c_void_p(st.cuda_stream))
    def wgrad(st): call("danhip_conv2d_bwd_weight", ctypes.byref(d), ptr(x), ptr(dy), ptr(dw), ptr(db), shp[3], ctypes.c_void_p(st.cuda_stream))
    def seq():
        dgrad(s1); wgrad(s1)
    def conc():
        dgrad(s1); wgrad(s2)
    res = []
    for fn in (seq, conc):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s1); s2.wait_event(e0)
        for _ in range(20): fn()
        e2 = torch.cuda.Event(); e2.record(s2); s1.wait_event(e2); e1.record(s1)
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1)/20)
    print("%-8s dgrad+wgrad sequential %.3f ms, two streams %.3f ms" % (name, res[0], res[1]))
