"""profiling build: the fused DCN forward with parts switched off (which role bounds a step)"""
import os, sys
os.environ["SIMPLEDET_AMD_LIB"] = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libsimpledet_ops_hip_prof.so")
sys.path.insert(0, ".")
import torch
from simpledet_amd import ops
from simpledet_amd._lib import lib
def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
torch.manual_seed(0)
N, C, H, W, F = 16, 256, 50, 84, 256
x = torch.randn(N, C, H, W, device="cuda"); off = torch.randn(N, 72, H, W, device="cuda") * 2; wt = torch.randn(F, C, 3, 3, device="cuda") * 0.05
for ab in (0, 31, 1, 2, 4, 8, 16):
    lib().set_tuning("dcn_fused_ablate", ab)
    print("ablate", ab, "(1 no sampling, 2 no MFMA, 4 no window loads, 8 no A loads, 16 no B reads): %.3f ms" % t(lambda: ops.deform_conv_forward(x, off, wt, 1, 1, 1, 4)))
