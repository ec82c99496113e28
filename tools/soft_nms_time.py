"""Scratch (GPU box): soft-NMS 1280 x 1000 boxes (the bench_ops problem), kernel time by events."""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from simpledet_amd import ops, synth
from simpledet_amd._lib import lib

P, N = 1280, 1000
dets = np.stack([synth.nms_dets(1000 + (p % 64), N) for p in range(P)])
t = torch.from_numpy(dets).cuda()
for T in (256, 128, 256):
    lib().set_tuning("soft_nms_threads", T)
    for _ in range(2):
        ops.soft_nms_batched(t, None, 0.5, 0.5, 0.001, 1)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        r = ops.soft_nms_batched(t, None, 0.5, 0.5, 0.001, 1)
    e1.record()
    torch.cuda.synchronize()
    print("threads %d: %.3f ms, mean kept %.1f" % (T, e0.elapsed_time(e1) / 5, float(r[2].float().mean())))
