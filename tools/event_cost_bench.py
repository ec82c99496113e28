"""Scratch experiment (GPU box): what in bench.py's timed step costs the ~18 us the bare step does not have?"""
import sys
import time

import torch

sys.path.insert(0, ".")
from simpledet_amd import ops, synth  # noqa: E402
from simpledet_amd._lib import lib  # noqa: E402

strides = list(synth.FPN_STRIDES)
feats = [torch.from_numpy(f).cuda() for f in synth.feature_maps(0, 2, 256, synth.FPN_SHAPES)]
rois = torch.from_numpy(synth.random_rois(0, 2, 512)).cuda()
dy = torch.randn((2, 512, 256, 7, 7), device="cuda")
d_feats = [torch.empty_like(f) for f in feats]
N = 200
evs = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(N + 40)]


def step(i, nev, dispatch):
    e = evs[i]
    if nev >= 1:
        e[0].record()
    out, am = ops.fpn_roi_align_forward_packed(feats, rois, strides, (7, 7), plan=True)
    if nev >= 3:
        e[1].record()
    if dispatch:
        lib().cdll.sd_last_dispatch()
    ops.fpn_roi_align_backward_packed(dy, rois, am, None, strides, d_feats=d_feats)
    if nev >= 2:
        e[2].record()
    if dispatch:
        lib().cdll.sd_last_dispatch()


def timeit(nev, dispatch):
    for i in range(20):
        step(i, nev, dispatch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(N):
        step(20 + i, nev, dispatch)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / N


t_pre = time.perf_counter()
while time.perf_counter() - t_pre < 0.3:
    step(0, 0, False)
torch.cuda.synchronize()
for rep in range(2):
    print("bare %.4f | dispatch-name calls only %.4f | 1 event/step %.4f | 2 events/step %.4f | 3 events/step %.4f | 3 events + names %.4f"
          % (timeit(0, False), timeit(0, True), timeit(1, False), timeit(2, False), timeit(3, False), timeit(3, True)))
