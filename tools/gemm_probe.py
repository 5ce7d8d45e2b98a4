import torch, time
dev = torch.device("cuda:0")
def bench(M, K, N, tag):
    a = torch.randn((M, K), device=dev, dtype=torch.bfloat16)
    b = torch.randn((K, N), device=dev, dtype=torch.bfloat16)
    bt = torch.randn((N, K), device=dev, dtype=torch.bfloat16)
    for name, fn in (("A@B", lambda: a @ b), ("A@Bt.T", lambda: a @ bt.t()), ("At.T@C (wgrad)", None)):
        if fn is None:
            c = torch.randn((M, N), device=dev, dtype=torch.bfloat16)
            fn = lambda: a.t() @ c
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("%-8s %-16s M=%d K=%d N=%d  %.3f ms  %.0f TFLOP/s" % (tag, name, M, K, N, ms, 2.0 * M * K * N / ms / 1e9))
bench(409600, 2304, 256, "dfm160")
bench(409600, 256, 2304, "dfm160dg")
bench(409600, 256, 256, "lat160")
bench(409600, 256, 1024, "up1024")
bench(102400, 2304, 256, "dfm80")
bench(6400, 1024, 1024, "fc7")
