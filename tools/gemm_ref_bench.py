"""Reference point for the fp32 MFMA GEMM of the DCN layer: the same three products through
torch.bmm (rocBLAS / hipBLASLt, fp32) and through sd_gemm_f32, on the (16,256,50,84) layer."""
import sys, time
sys.path.insert(0, ".")
import torch
from simpledet_amd import ops

def t(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it

N, F, K, P = 16, 256, 2304, 4200
w = torch.randn(1, F, K, device="cuda").expand(N, F, K).contiguous()
col = torch.randn(N, K, P, device="cuda")
dy = torch.randn(N, F, P, device="cuda")
fl = 2.0 * N * F * K * P
for name, f_t, f_s in [
    ("y = W col        ", lambda: torch.bmm(w, col), lambda: ops.gemm_f32(w, col)),
    ("dcol = W^T dY    ", lambda: torch.bmm(w.transpose(1, 2), dy), lambda: ops.gemm_f32(w, dy, trans_a=True)),
    ("dW_n = dY col^T  ", lambda: torch.bmm(dy, col.transpose(1, 2)), lambda: ops.gemm_f32(dy, col, trans_b=True)),
]:
    a, b = t(f_t), t(f_s)
    print("%s torch %.3f ms %.1f TF | sd_gemm_f32 %.3f ms %.1f TF" % (name, a, fl / a / 1e9, b, fl / b / 1e9))

# tile-width A/B (sd_gemm_f32 only)
from simpledet_amd._lib import lib
for J, bk in ((1, 16), (2, 16), (3, 16), (0, 16), (1, 32), (2, 32)):
    r = []
    for f_s in (lambda: ops.gemm_f32(w, col), lambda: ops.gemm_f32(w, dy, trans_a=True),
                lambda: ops.gemm_f32(dy, col, trans_b=True)):
        r.append(fl / t(f_s) / 1e9)
    print("J=%d (0 = auto) BK=%d: %.1f / %.1f / %.1f TF" % (J, bk, r[0], r[1], r[2]))
