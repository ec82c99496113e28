import sys, os
sys.path.insert(0, ".")
import numpy as np, torch
from simpledet_amd import ops
torch.manual_seed(0)
N, C, H, W, F = 2, 256, 50, 84, 256
x = torch.randn(N, C, H, W, device="cuda"); off = torch.randn(N, 72, H, W, device="cuda") * 2; wt = torch.randn(F, C, 3, 3, device="cuda") * 0.05
yu, _ = ops.deform_conv_forward(x, off, wt, 1, 1, 1, 4, keep_col=True)
bad = 0
for it in range(30):
    y = ops.deform_conv_forward(x, off, wt, 1, 1, 1, 4)
    e = float((y - yu).abs().max())
    if e > 1e-3:
        bad += 1
        idx = ((y - yu).abs() > 1e-3).nonzero()
        print("iter", it, "err", e, "n bad", idx.shape[0], "first", idx[:3].tolist())
print(os.environ.get("SIMPLEDET_AMD_LIB", "default lib"), "bad runs:", bad, "of 30")
