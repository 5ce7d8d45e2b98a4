import torch, time
dev = torch.device("cuda:0")
n = 16*640*640*64
x = torch.randn(n, device=dev, dtype=torch.float32).to(torch.bfloat16)
y = torch.empty_like(x)
def t(f, it=20):
    for _ in range(3): f()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/it
ms = t(lambda: y.zero_()); print("fill 839MB  %.3f ms  %.2f TB/s write" % (ms, n*2/ms/1e9))
ms = t(lambda: y.copy_(x)); print("copy 839MB  %.3f ms  %.2f TB/s r+w" % (ms, 2*n*2/ms/1e9))
ms = t(lambda: x.sum()); print("read 839MB  %.3f ms  %.2f TB/s read" % (ms, n*2/ms/1e9))
