#!/usr/bin/env python
"""Per-wave phase clocks of the split-bf16 GEMM (profiling build only): wait for the prefetched
global loads / split + LDS store + barriers / fragment reads + MFMAs, per k step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("SIMPLEDET_AMD_LIB", os.path.join(ROOT, "tools", "libsimpledet_ops_hip_prof.so"))
import numpy as np
import torch
from simpledet_amd import ops
from simpledet_amd._lib import lib

N, F, K, P = 16, 256, 2304, 4200
w = torch.randn(1, F, K, device="cuda").expand(N, F, K).contiguous()
col = torch.randn(N, K, P, device="cuda")
dy = torch.randn(N, F, P, device="cuda")
cases = [("y = W col", lambda: ops.gemm_f32(w, col)), ("dcol = W^T dY", lambda: ops.gemm_f32(w, dy, trans_a=True)),
         ("dW = dY col^T", lambda: ops.gemm_f32(dy, col, trans_b=True))]
cap = 16 * 1024 * 4
abl = int(sys.argv[1]) if len(sys.argv) > 1 else 0
lib().set_tuning("gemm_ablate", abl)
print("ablate =", abl, "(1: no A prefetch, 2: no B prefetch; results are wrong, timing only)")
def t_ms(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for name, fn in cases:
    print("%s: %.3f ms" % (name, t_ms(fn)))
for name, fn in cases:
    fn(); fn()
    dbg = torch.zeros(cap * 8, dtype=torch.int64, device="cuda")
    p = dbg.data_ptr()
    lo = p & 0xffffffff
    lib().set_tuning("roi_align_dbg_lo", lo - (1 << 32) if lo & 0x80000000 else lo)
    lib().set_tuning("roi_align_dbg_hi", p >> 32)
    lib().set_tuning("gemm_dbg_cap", cap)
    fn()
    torch.cuda.synchronize()
    lib().set_tuning("roi_align_dbg_lo", 0)
    lib().set_tuning("roi_align_dbg_hi", 0)
    d = dbg.cpu().numpy().reshape(cap, 8)
    full = d
    d = d[d[:, 4] > 0]
    # timeline of XCD 0 (block b -> XCD b % 8; each XCD has its own clock): start / end of wave 0 of its blocks
    blk = full.reshape(-1, 4, 8)[:, 0, :]
    x0 = blk[0::8]
    x0 = x0[x0[:, 4] > 0]
    t0 = x0[:, 3].min()
    st = np.sort(x0[:, 3] - t0)
    en = np.sort(x0[:, 4] - t0)
    print("   XCD 0: %d blocks; starts (k ticks) p0 %d p25 %d p50 %d p75 %d p100 %d; ends p0 %d p50 %d p100 %d; "
          "concurrent at the median start: %d" % (len(x0), st[0] / 1e3, st[len(st) // 4] / 1e3, st[len(st) // 2] / 1e3,
          st[3 * len(st) // 4] / 1e3, st[-1] / 1e3, en[0] / 1e3, en[len(en) // 2] / 1e3, en[-1] / 1e3,
          ((x0[:, 3] - t0 <= st[len(st) // 2]) & (x0[:, 4] - t0 > st[len(st) // 2])).sum()))
    ks = d[:, 5].astype(float)
    span = d[:, 4].max() - d[:, 3].min()
    print("%s: %d waves, k steps %d, kernel span %d ticks; per k step (ticks): load wait %.0f  split+store+barriers %.0f  "
          "issue next loads %.0f  reads+MFMA %.0f;  wave lifetime mean %.0f (prologue+epilogue %.0f)" % (
              name, len(d), ks[0], span, (d[:, 0] / ks).mean(), (d[:, 1] / ks).mean(), (d[:, 6] / ks).mean(), (d[:, 2] / ks).mean(),
              (d[:, 4] - d[:, 3]).mean(), ((d[:, 4] - d[:, 3]) - d[:, :3].sum(1) - d[:, 6]).mean()))
