"""Scratch experiment (GPU box): the bench step (merged pre-pass + forward + backward = 3 launches) issued
launch by launch against the same three launches replayed from ONE captured HIP graph.
   python tools/graph_step_bench.py [steps]"""
import sys
import time

import torch

sys.path.insert(0, ".")
from simpledet_amd import ops, synth  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
strides = list(synth.FPN_STRIDES)
feats = [torch.from_numpy(f).cuda() for f in synth.feature_maps(0, 2, 256, synth.FPN_SHAPES)]
rois = torch.from_numpy(synth.random_rois(0, 2, 512)).cuda()
dy = torch.randn((2, 512, 256, 7, 7), device="cuda")
d_feats = [torch.empty_like(f) for f in feats]


def step():
    out, am = ops.fpn_roi_align_forward_packed(feats, rois, strides, (7, 7), plan=True)
    ops.fpn_roi_align_backward_packed(dy, rois, am, None, strides, d_feats=d_feats)
    return out


def timeit(fn, n):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / n


t_pre = time.perf_counter()
while time.perf_counter() - t_pre < 0.3:
    step()
torch.cuda.synchronize()
eager = [timeit(step, steps) for _ in range(3)]
ref = [g.clone() for g in d_feats]
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    out = step()
for t in d_feats:
    t.zero_()
g.replay()
torch.cuda.synchronize()
same = all(torch.equal(a, b) for a, b in zip(ref, d_feats))
graph = [timeit(g.replay, steps) for _ in range(3)]
eager2 = [timeit(step, steps) for _ in range(3)]
print("eager ms/step", ["%.4f" % v for v in eager], "after", ["%.4f" % v for v in eager2])
print("graph ms/step", ["%.4f" % v for v in graph], "same results:", same)
