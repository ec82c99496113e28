"""time the backward's data-gradient half only (dcol product + coord + col2im) for knob sets: kernel times via tools/kt_cmd.sh"""
import sys
sys.path.insert(0, ".")
import torch
from simpledet_amd import ops
from simpledet_amd._lib import lib
torch.manual_seed(0)
N, C, H, W, F = 16, 256, 50, 84, 256
x = torch.randn(N, C, H, W, device="cuda"); wt = torch.randn(F, C, 3, 3, device="cuda") * 0.05
dy = torch.randn(N, F, H, W, device="cuda")
off = torch.randn(N, 72, H, W, device="cuda") * 2.0
grads = (torch.empty_like(x), torch.empty(N, 72, H, W, device="cuda"), torch.empty_like(wt))
for a in sys.argv[1:]:
    for kv in a.split(","):
        k, v = kv.split("="); lib().set_tuning(k, int(v))
    for _ in range(6):
        ops.deform_conv_backward(dy, x, off, wt, 1, 1, 1, 4, grads=grads, req=("write", "write", "null"))
    torch.cuda.synchronize()
