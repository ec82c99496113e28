"""A/B of GEMM tuning knobs on the DCN layer's backward (one box, alternating): python tools/dcn_gemm_ab.py key=v[,key=v] ..."""
import sys
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
x = torch.randn(N, C, H, W, device="cuda"); wt = torch.randn(F, C, 3, 3, device="cuda") * 0.05
dy = torch.randn(N, F, H, W, device="cuda")
off = torch.randn(N, 72, H, W, device="cuda") * 2.0
grads = (torch.empty_like(x), torch.empty(N, 72, H, W, device="cuda"), torch.empty_like(wt))
_, fws = ops.deform_conv_forward(x, off, wt, 1, 1, 1, 4, keep_col=True)
settings = [dict(kv.split("=") for kv in a.split(",") if kv) for a in sys.argv[1:]] or [{}]
ref = None
for rep in range(2):
    for s in settings:
        for k, v in s.items(): lib().set_tuning(k, int(v))
        b = t(lambda: ops.deform_conv_backward(dy, x, off, wt, 1, 1, 1, 4, grads=grads))
        bc = t(lambda: ops.deform_conv_backward(dy, x, off, wt, 1, 1, 1, 4, grads=grads, fwd_ws=fws))
        fu = t(lambda: ops.deform_conv_forward(x, off, wt, 1, 1, 1, 4, keep_col=True))
        g = [v.clone() for v in grads]
        if ref is None: ref = g
        d = [float((a - b_).abs().max() / b_.abs().max()) for a, b_ in zip(g, ref)]
        print("%-50s bwd %.3f  bwd(col kept) %.3f  fwd im2col+gemm %.3f   rel diff to first dX %.1e dOff %.1e dW %.1e" % (s, b, bc, fu, *d), flush=True)
