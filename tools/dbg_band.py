import sys, numpy as np, torch
sys.path.insert(0, '.')
from simpledet_amd import ops, synth
from oracle import pyoracle as orc
STRIDES = list(synth.FPN_STRIDES)
feats = synth.feature_maps(0, batch=2, channels=16)
rois = synth.random_rois(1, 2, 64)
want = orc.fpn_roi_align_fwd(feats, rois, STRIDES, (7, 7), nthreads=8)
tf = [torch.from_numpy(f).cuda() for f in feats]
got = ops.fpn_roi_align_forward(tf, torch.from_numpy(rois).cuda(), STRIDES, (7, 7))
g = got[0].cpu().numpy(); w = want[0]
bad = (g != w) & ~(np.isnan(g) & np.isnan(w))
lv = ops.fpn_roi_assign(torch.from_numpy(rois).cuda(), STRIDES)[1].cpu().numpy().reshape(2, 64)
print("bad total", bad.sum(), "of", bad.size)
for l in range(4):
    m = lv == l
    print("level", l, "rois", m.sum(), "bad", bad[m].sum())
b, r, c, p, q = np.nonzero(bad)
print("by image", np.bincount(b, minlength=2))
print("by channel", np.bincount(c, minlength=16))
print("by p", np.bincount(p, minlength=7))
print("by q", np.bincount(q, minlength=7))
rr = sorted(set(zip(b.tolist(), r.tolist())))
print("bad rois", len(rr), rr[:20])
for (bi, ri) in rr[:6]:
    print(bi, ri, rois[bi, ri], "lvl", lv[bi, ri], "bad bins", bad[bi, ri].sum(), "channels", sorted(set(np.nonzero(bad[bi, ri])[0].tolist())), "p", sorted(set(np.nonzero(bad[bi, ri])[1].tolist())))
