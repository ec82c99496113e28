"""DCN layer backward at several offset magnitudes (kernel-level numbers: run under tools/kt_summary or rocprofv3)"""
import sys
sys.path.insert(0, ".")
import torch
from simpledet_amd import ops
def t(fn, it=8):
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
grads = (torch.empty_like(x), torch.empty(N, 72, H, W, device="cuda"), torch.empty_like(wt))
for sc in (2.0, 1.0, 0.5, 0.0):
    off = torch.randn(N, 72, H, W, device="cuda") * sc
    col = ops.deform_im2col(x, off, (3, 3), 1, 1, 1, 4)
    print("sigma %.1f: backward %.3f ms   col2im alone (float adds) %.3f  coord %.3f  im2col %.3f" % (
        sc, t(lambda: ops.deform_conv_backward(dy, x, off, wt, 1, 1, 1, 4, grads=grads)),
        t(lambda: ops.deform_col2im(col, off, x.shape, (3, 3), 1, 1, 1, 4)),
        t(lambda: ops.deform_col2im_coord(col, x, off, (3, 3), 1, 1, 1, 4)),
        t(lambda: ops.deform_im2col(x, off, (3, 3), 1, 1, 1, 4))))
    del col
