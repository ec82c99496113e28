#!/usr/bin/env python
"""Per-wave phase clocks of the resident forward workgroups (profiling build only).
usage: SIMPLEDET_AMD_LIB=tools/libsimpledet_ops_hip_prof.so python tools/fwd_phase_clocks.py [key=value ...]"""
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
nblk = 4096
dbg = torch.zeros(nblk * 16 * 8, dtype=torch.int64, device="cuda")
p = dbg.data_ptr()
lib().set_tuning("roi_align_dbg_lo", (p & 0xffffffff) - (1 << 32) if (p & 0x80000000) else p & 0xffffffff)
lib().set_tuning("roi_align_dbg_hi", p >> 32)
ops.fpn_roi_align_forward_packed(feats, rois, strides, (7, 7))
torch.cuda.synchronize()
lib().set_tuning("roi_align_dbg_lo", 0)
lib().set_tuning("roi_align_dbg_hi", 0)
d = dbg.cpu().numpy().reshape(nblk, 16, 8)
# s_memtime / readcyclecounter ticks at 100 MHz on gfx9 (constant clock): report in us
tick_us = 1.0 / 100.0
used = d[:, :, 4].max(1) > 0
for lv in sorted(set(d[used][:, 0, 5].tolist())):
    m = used & (d[:, 0, 5] == lv)
    b = d[m]
    print("level %d: %d workgroups, G=%d, RoIs/WG mean %.1f" % (lv, m.sum(), b[0, 0, 7], b[:, 0, 6].mean()))
    print("   fill+list  %.2f us   table (per wave, sum) %.2f us   bins %.2f us   units/wave %.2f   wave total %.2f us  (max %.2f)" % (
        b[:, :, 0].mean() * tick_us, b[:, :, 1].mean() * tick_us, b[:, :, 2].mean() * tick_us,
        b[:, :, 3].mean(), b[:, :, 4].mean() * tick_us, b[:, :, 4].max() * tick_us))
    nu = np.maximum(b[:, :, 3], 1)
    print("   per unit: table %.2f us, bins %.2f us" % ((b[:, :, 1] / nu)[b[:, :, 3] > 0].mean() * tick_us,
                                                        (b[:, :, 2] / nu)[b[:, :, 3] > 0].mean() * tick_us))
