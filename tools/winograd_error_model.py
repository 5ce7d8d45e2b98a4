"""Winograd F(2x2, 3x3) on 16-bit MFMA operands: how much accuracy does it cost?  (kill criterion "error > 2x the direct
kernel's against oracle/tf_ops.conv2d_same on bf16-rounded inputs".)  CPU emulation, no GPU needed:

  direct   : x, w rounded to 16 bits; products exact, fp32 accumulation (what conv3x3_halo_kernel does); output rounded to 16 bits
  winograd : the same x, w; U = G g G^T and V = B^T d B formed in fp32 and ROUNDED TO 16 BITS (they are the MFMA operands);
             M = sum_c U . V in fp32; Y = A^T M A in fp32; output rounded to 16 bits
  exact    : float64 convolution of the rounded x, w.

Prints max / rms error of both against exact, in units of the output's max, before and after the final 16-bit rounding.
Usage: python tools/winograd_error_model.py [bf16|fp16] [Cin] [Cout] [HW]"""
import sys

import torch

dt = {"bf16": torch.bfloat16, "fp16": torch.float16}[sys.argv[1] if len(sys.argv) > 1 else "bf16"]
Cin = int(sys.argv[2]) if len(sys.argv) > 2 else 256
Cout = int(sys.argv[3]) if len(sys.argv) > 3 else 256
HW = int(sys.argv[4]) if len(sys.argv) > 4 else 32
torch.manual_seed(0)
r16 = lambda t: t.to(dt).to(torch.float32)
x = r16(torch.relu(torch.randn(1, Cin, HW, HW)))                      # post-ReLU activations
w = r16(torch.randn(Cout, Cin, 3, 3) * (2.0 / (9 * Cin)) ** 0.5)
exact = torch.nn.functional.conv2d(x.double(), w.double(), padding=1)
direct32 = torch.nn.functional.conv2d(x, w, padding=1)

G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
Bt = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
At = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)
U = r16(torch.einsum("ai,ocij,bj->ocab", G, w, G))                     # [Co, Ci, 4, 4]
xp = torch.nn.functional.pad(x, (1, 1, 1, 1))
T = HW // 2
tiles = xp.unfold(2, 4, 2).unfold(3, 4, 2)                             # [1, Ci, T, T, 4, 4]
V = r16(torch.einsum("ai,nctuij,bj->nctuab", Bt, tiles, Bt))
M = torch.einsum("ocab,nctuab->notuab", U, V)                          # fp32 accumulation over channels, per position
Y = torch.einsum("ia,notuab,jb->notuij", At, M, At)                    # [1, Co, T, T, 2, 2]
wino32 = Y.permute(0, 1, 2, 4, 3, 5).reshape(1, Cout, HW, HW)

scale = exact.abs().max().item()
def report(name, y):
    e = (y.double() - exact).abs()
    print("%-28s max %.3e  rms %.3e   (of the output's max %.3f)" % (name, e.max().item() / scale, e.pow(2).mean().sqrt().item() / scale, scale))
    return e.max().item() / scale, e.pow(2).mean().sqrt().item() / scale
print("dtype %s  Cin %d  Cout %d  map %dx%d" % (sys.argv[1] if len(sys.argv) > 1 else "bf16", Cin, Cout, HW, HW))
d0 = report("direct, fp32 accumulators", direct32)
w0 = report("winograd, fp32 accumulators", wino32)
d1 = report("direct, stored 16-bit", r16(direct32))
w1 = report("winograd, stored 16-bit", r16(wino32))
print("ratio before the store: max %.1fx rms %.1fx;  after the 16-bit store: max %.2fx rms %.2fx" % (w0[0] / d0[0], w0[1] / d0[1], w1[0] / d1[0], w1[1] / d1[1]))
