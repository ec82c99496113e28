"""the un-fused FPN graph's forward (assign -> 4 x ROIAlign_v2 -> add_n) and its four level ops alone, by events"""
import sys
sys.path.insert(0, ".")
import torch, numpy as np
from simpledet_amd import ops
from simpledet_amd import synth
torch.manual_seed(0)
strides = list(synth.FPN_STRIDES)
feats = [torch.from_numpy(f).cuda() for f in synth.feature_maps(0, 2, 256, synth.FPN_SHAPES)]
rois = torch.from_numpy(synth.random_rois(0, 2, 512)).cuda()
def t(fn, it=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
per, _ = ops.fpn_roi_assign(rois, strides)
def levels():
    return [ops.roi_align_v2_forward(f, p, (7, 7), 1.0 / s)[0] for f, p, s in zip(feats, per, strides)]
def graph():
    per2, _ = ops.fpn_roi_assign(rois, strides)
    tot = None
    for f, p, s in zip(feats, per2, strides):
        o = ops.roi_align_v2_forward(f, p, (7, 7), 1.0 / s)[0]
        tot = o if tot is None else tot + o
    return tot
print("void rows per level:", [int((p.abs().sum(-1) == 0).sum()) for p in per])
for rep in range(3):
    print("four level ops %.3f ms   whole graph %.3f ms   per level %s" % (
        t(levels), t(graph), ["%.3f" % t(lambda f=f, p=p, s=s: ops.roi_align_v2_forward(f, p, (7, 7), 1.0 / s)) for f, p, s in zip(feats, per, strides)]))
