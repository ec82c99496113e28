"""dcol = W^T dY: A given as W (F x K: transposed read) against A given as W^T (K x F: reduction index contiguous)"""
import sys
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
torch.manual_seed(0)
N, F, K, P = 16, 256, 2304, 4200
w = torch.randn(N, F, K, device="cuda") * 0.05      # (Bt, F, K): op(A) = A^T
wt = w.transpose(1, 2).contiguous()                   # (Bt, K, F)
dy = torch.randn(N, F, P, device="cuda")
out = torch.empty(N, K, P, device="cuda")
print("A = W, trans_a   : %.3f ms" % t(lambda: ops.gemm_f32(w, dy, trans_a=True, out=out)))
print("A = W^T stored   : %.3f ms" % t(lambda: ops.gemm_f32(wt, dy, out=out)))
dyt = dy.transpose(1, 2).contiguous()                 # (Bt, P, F): B^T stored
print("A = W^T, B^T stored: %.3f ms" % t(lambda: ops.gemm_f32(wt, dyt, trans_b=True, out=out)))
