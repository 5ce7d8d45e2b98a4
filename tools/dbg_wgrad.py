import sys, torch
sys.path.insert(0, '/root/repo')
from dan_amd import ops
dev = torch.device('cuda:0')
C = int(sys.argv[1]) if len(sys.argv) > 1 else 64
for (p0, c0) in [(0, 0), (1, 0), (0, 1), (5, 9), (37, 21), (63, 63)]:
    x = torch.zeros((1, 8, 8, C), dtype=torch.bfloat16, device=dev)
    x.view(64, C)[p0, c0] = 1
    xd = x.requires_grad_(True)
    w = torch.zeros((1, 1, C, C), device=dev, requires_grad=True)
    y = ops.conv2d(xd, w, None, stride=1, relu=False)
    dy = torch.zeros((64, C), dtype=torch.bfloat16, device=dev)
    dy[p0] = torch.arange(1, C + 1, device=dev).to(torch.bfloat16)
    y.backward(dy.view(1, 8, 8, C))
    g = w.grad.view(C, C).cpu()
    nz = g.nonzero()
    print('p0', p0, 'c0', c0, 'rows with nz:', sorted(set(nz[:, 0].tolist()))[:8], 'n nz', len(nz))
    if len(nz):
        r = nz[0, 0].item()
        print('   row', r, 'vals', g[r, :20].tolist())
