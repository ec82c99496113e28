"""sd_gemm_f32 on the bf16 matrix cores (hi/lo split, three MFMA terms) against the fp32 MFMA
path: error against an fp64 product and time, on the three products of the (16,256,50,84) DCN layer
and on ragged shapes in all four layouts."""
import sys
sys.path.insert(0, ".")
import numpy as np
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
for shape in [(1, 128, 128, 16), (2, 256, 300, 72), (3, 70, 4200, 33), (1, 1, 1, 1), (2, 129, 257, 17),
              (2, 256, 4200, 2304)]:
    Bt, M, N, K = shape
    A = torch.randn(Bt, M, K, device="cuda")
    B = torch.randn(Bt, K, N, device="cuda")
    want = torch.bmm(A.double(), B.double())
    for split in (0, 1):
        lib().set_tuning("deform_gemm_split", split)
        errs = []
        for ta in (False, True):
            for tb in (False, True):
                a = A.transpose(1, 2).contiguous() if ta else A
                b = B.transpose(1, 2).contiguous() if tb else B
                got = ops.gemm_f32(a, b, ta, tb)
                errs.append(float((got.double() - want).abs().max() / want.abs().max()))
        c0 = torch.randn(Bt, M, N, device="cuda")
        got = ops.gemm_f32(A, B, out=c0.clone(), accumulate=1)
        errs.append(float((got.double() - want - c0.double()).abs().max() / want.abs().max()))
        got = ops.gemm_f32(A, B, out=c0.clone(), accumulate=2)
        errs.append(float((got.double() - want - c0.double()).abs().max() / want.abs().max()))
        print(shape, "split", split, "max err / max|C| per layout:", " ".join("%.1e" % e for e in errs))

N, F, K, P = 16, 256, 2304, 4200
w = torch.randn(1, F, K, device="cuda").expand(N, F, K).contiguous()
col = torch.randn(N, K, P, device="cuda")
dy = torch.randn(N, F, P, device="cuda")
fl = 2.0 * N * F * K * P
lib().set_tuning("deform_gemm_split", 1)
for nm, got, want in (("y", ops.gemm_f32(w, col), torch.bmm(w.double(), col.double())),
                      ("dcol", ops.gemm_f32(w, dy, trans_a=True), torch.bmm(w.double().transpose(1, 2), dy.double())),
                      ("dW", ops.gemm_f32(dy, col, trans_b=True), torch.bmm(dy.double(), col.double().transpose(1, 2)))):
    print("DCN shape %s: max err / max|C| = %.2e" % (nm, float((got.double() - want).abs().max() / want.abs().max())))
    del got, want
acc = torch.zeros(1, F, K, device="cuda")
ops.gemm_f32(dy, col, trans_b=True, out=acc.expand(N, F, K), accumulate=2) if False else None
for split in (0, 1, 0, 1):
    lib().set_tuning("deform_gemm_split", split)
    r = []
    for f_s in (lambda: ops.gemm_f32(w, col), lambda: ops.gemm_f32(w, dy, trans_a=True),
                lambda: ops.gemm_f32(dy, col, trans_b=True)):
        r.append(t(f_s))
    print("split=%d: y=W col %.3f ms (%.0f TF-eq) | dcol=W^T dY %.3f ms (%.0f) | dW=dY col^T %.3f ms (%.0f)"
          % (split, r[0], fl / r[0] / 1e9, r[1], fl / r[1] / 1e9, r[2], fl / r[2] / 1e9))
