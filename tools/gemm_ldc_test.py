"""dcol-shaped product (M 2304, N 4200, K 256, batch 16) with the natural and a line-aligned row pitch of C"""
import sys, ctypes
sys.path.insert(0, ".")
import torch
from simpledet_amd._lib import lib
from simpledet_amd.ops import _p, _stream
def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
torch.manual_seed(0)
B, M, N, K = 16, 2304, 4200, 256
w = torch.randn(K, M, device="cuda") * 0.05        # W[f][(c,tap)]: A^T
dy = torch.randn(B, K, N, device="cuda")
ws = torch.empty(64, device="cuda", dtype=torch.uint8)
for nt in (0, 1):
    lib().set_tuning("deform_gemm_nt", nt)
    for ldc in (4200, 4224, 4352):
        c = torch.empty(B, M, ldc, device="cuda")
        f = lambda: lib().call("sd_gemm_f32_ws", 1, 0, M, N, K, _p(w), M, 0, _p(dy), N, K * N, _p(c), ldc, M * ldc, B, 0,
                               _p(ws), ctypes.c_size_t(64), _stream())
        print("nt %d ldc %d: %.3f ms" % (nt, ldc, t(f)), flush=True)
