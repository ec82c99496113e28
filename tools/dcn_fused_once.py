"""a few calls of the fused DCN forward (for rocprofv3 --kernel-trace --stats)"""
import sys
sys.path.insert(0, ".")
import torch
from simpledet_amd import ops
torch.manual_seed(0)
N, C, H, W, F = 16, 256, 50, 84, 256
sc = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
x = torch.randn(N, C, H, W, device="cuda"); off = torch.randn(N, 72, H, W, device="cuda") * sc; wt = torch.randn(F, C, 3, 3, device="cuda") * 0.05
for _ in range(20):
    y = ops.deform_conv_forward(x, off, wt, 1, 1, 1, 4)
torch.cuda.synchronize()
