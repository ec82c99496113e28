#!/usr/bin/env python
"""Per-wave phase clocks of the band-resident forward workgroups (profiling build only).
usage: python tools/fwd_phase_clocks.py [key=value ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SIMPLEDET_AMD_LIB", os.path.join(ROOT, "tools", "libsimpledet_ops_hip_prof.so"))
import numpy as np
import torch
from simpledet_amd import ops, synth
from simpledet_amd._lib import lib

for kv in sys.argv[1:]:
    k, v = kv.split("=")
    lib().set_tuning(k, int(v))
feats = [torch.from_numpy(f).cuda() for f in synth.feature_maps(0, 2, 256)]
rois = torch.from_numpy(synth.random_rois(0, 2, 512)).cuda()
strides = list(synth.FPN_STRIDES)
for _ in range(3):
    ops.fpn_roi_align_forward_packed(feats, rois, strides, (7, 7))
nblk = 256
dbg = torch.zeros(nblk * 16 * 8 + nblk * 4 * 8, dtype=torch.int64, device="cuda")
p = dbg.data_ptr()
lo = p & 0xffffffff
lib().set_tuning("roi_align_dbg_lo", lo - (1 << 32) if lo & 0x80000000 else lo)
lib().set_tuning("roi_align_dbg_hi", p >> 32)
ops.fpn_roi_align_forward_packed(feats, rois, strides, (7, 7))
torch.cuda.synchronize()
lib().set_tuning("roi_align_dbg_lo", 0)
lib().set_tuning("roi_align_dbg_hi", 0)
raw = dbg.cpu().numpy()
d = raw[:nblk * 16 * 8].reshape(nblk, 16, 8)
used = d[:, :, 4].max(1) > 0
t0 = d[used][:, :, 7].min()
print("workgroups that ran:", used.sum(), " span (ticks) %d" % ((d[used][:, :, 7] + d[used][:, :, 4]).max() - t0))
b = d[used]
print("per wave (ticks): setup %.0f  barrier wait %.0f  compute %.0f  total mean %.0f  min %d  max %d" % (
    b[:, :, 0].mean(), b[:, :, 1].mean(), b[:, :, 2].mean(), b[:, :, 4].mean(), b[:, :, 4].min(), b[:, :, 4].max()))

print("partition before the first visit (ticks): mean %.0f max %d;  outside the visits: mean %.0f" % (
    b[:, 0, 5].mean(), b[:, 0, 5].max(), (b[:, :, 4].max(1) - (b[:, 0, 0] + b[:, 0, 1] + b[:, 0, 2])).mean()))
print("units per workgroup: mean %.2f max %d; items visited per workgroup mean %.0f" % (b[:, 0, 6].mean(), b[:, 0, 6].max(), b[:, 0, 3].mean()))
tot = b[:, :, 4].max(1)
inv = b[:, 0, 0] + b[:, 0, 1] + b[:, 0, 2]
outv = b[:, :, 4].max(1) - inv
print("inside the visits (wave 0): p10 %d p50 %d p90 %d max %d;  outside: p10 %d p50 %d p90 %d max %d" % (
    tuple(np.percentile(inv, [10, 50, 90, 100]).astype(int)) + tuple(np.percentile(outv, [10, 50, 90, 100]).astype(int))))
print("start tick spread (t_begin - min): p50 %d p90 %d max %d" % tuple(np.percentile(b[:, 0, 7] - b[:, 0, 7].min(), [50, 90, 100]).astype(int)))
print("visits per workgroup histogram:", np.bincount(b[:, 0, 6].astype(int)).tolist())
print("workgroup total ticks: p10 %d p50 %d p90 %d max %d" % tuple(np.percentile(tot, [10, 50, 90, 100]).astype(int)))

v = raw[nblk * 16 * 8:].reshape(nblk, 4, 8)
v = v[v[:, :, 4] == 1]
for lv in sorted(set(v[:, 0].tolist())):
    m = v[:, 0] == lv
    w = v[m]
    dt = (w[:, 5] - w[:, 3]).astype(float)
    print("level %d: %d visits, fills/visit mean %.1f, items mean %.0f, ticks per fill: mean %.0f  (visits with >= 4 fills: %.0f)" % (
        lv, m.sum(), w[:, 1].mean(), w[:, 2].mean(), (dt / w[:, 1]).mean(),
        (dt / w[:, 1])[w[:, 1] >= 4].mean() if (w[:, 1] >= 4).any() else float("nan")))
    # least squares: ticks = setup + fills * per_fill
    A = np.stack([np.ones(len(w)), w[:, 1].astype(float)], 1)
    sol = np.linalg.lstsq(A, dt, rcond=None)[0]
    print("     fit: setup %.0f + %.0f per fill" % (sol[0], sol[1]))
out = os.path.join(ROOT, "gpurun_out")
os.makedirs(out, exist_ok=True)
np.save(os.path.join(out, "fwd_visits.npy"), raw[nblk * 16 * 8:].reshape(nblk, 4, 8))
np.save(os.path.join(out, "fwd_waves.npy"), d)
