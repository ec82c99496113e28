"""profiling build: per-wave clocks of the fused DCN forward (total, waiting at the step barrier, set-up)"""
import os, sys
os.environ["SIMPLEDET_AMD_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsimpledet_ops_hip_prof.so")
sys.path.insert(0, ".")
import numpy as np, torch
from simpledet_amd import ops
from simpledet_amd._lib import lib
torch.manual_seed(0)
N, C, H, W, F = 16, 256, 50, 84, 256
sc = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
x = torch.randn(N, C, H, W, device="cuda"); off = torch.randn(N, 72, H, W, device="cuda") * sc; wt = torch.randn(F, C, 3, 3, device="cuda") * 0.05
for _ in range(3): ops.deform_conv_forward(x, off, wt, 1, 1, 1, 4)
nblk = 768
dbg = torch.zeros(nblk * 8 * 4, dtype=torch.int64, device="cuda")
p = dbg.data_ptr(); lo = p & 0xffffffff
lib().set_tuning("roi_align_dbg_lo", lo - (1 << 32) if lo & 0x80000000 else lo)
lib().set_tuning("roi_align_dbg_hi", p >> 32)
ops.deform_conv_forward(x, off, wt, 1, 1, 1, 4); torch.cuda.synchronize()
lib().set_tuning("roi_align_dbg_lo", 0); lib().set_tuning("roi_align_dbg_hi", 0)
d = dbg.cpu().numpy().reshape(nblk, 8, 4)
d = d[d[:, 0, 0] > 0]
print("blocks", d.shape[0], "(100 MHz counter ticks: x24 = core clocks)")
for name, ws in (("matrix waves", [0, 1, 2, 3]), ("sampling waves", [4, 5, 6]), ("loader wave", [7])):
    t = d[:, ws, :]
    print("%-15s total %8.0f  barrier wait %8.0f (%.0f%%)  set-up %8.0f (%.0f%%)" % (
        name, t[..., 0].mean(), t[..., 1].mean(), 100 * t[..., 1].mean() / t[..., 0].mean(),
        t[..., 2].mean(), 100 * t[..., 2].mean() / t[..., 0].mean()))
