import sys
sys.path.insert(0, ".")
import torch, numpy as np
from simpledet_amd import ops, synth
def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
feats = [torch.randn((2, 256, h, w), device="cuda") for h, w in synth.FPN_SHAPES]
for pooled, num in (((14, 14), 128), ((7, 7), 512)):
    r = torch.from_numpy(synth.random_rois(0, 2, num)).cuda()
    s4 = [4, 8, 16, 32]
    f16 = [f.half() for f in feats]
    o16, am16 = ops.fpn_roi_align_forward_packed_f16(f16, r, s4, pooled)
    o32, am32 = ops.fpn_roi_align_forward_packed(feats, r, s4, pooled)
    dy16 = torch.randn_like(o16); dy32 = dy16.float()
    shapes = [f.shape for f in feats]
    g16 = [torch.empty_like(f) for f in f16]; g32 = [torch.empty_like(f) for f in feats]
    print(pooled, "bwd fp32 %.4f ms | fp16 native %.4f ms | fp16 via casts %.4f ms" % (
        t(lambda: ops.fpn_roi_align_backward_packed(dy32, r, am32, shapes, s4, d_feats=g32)),
        t(lambda: ops.fpn_roi_align_backward_packed_f16(dy16, r, am16, shapes, s4, d_feats=g16)),
        t(lambda: ops.fpn_roi_align_backward_packed_f16(dy16, r, am16, shapes, s4, d_feats=g16, native=False))))
