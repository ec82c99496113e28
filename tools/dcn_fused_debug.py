import sys
sys.path.insert(0, ".")
import numpy as np, torch
from simpledet_amd import ops
from tests.test_deform_conv import _case
cfg = dict(N=2, C=64, H=12, W=16, F=24, dg=4)
x, off, w, kw = _case(51, **cfg)
a = dict(pad=kw["pad"], stride=kw["stride"], dilate=kw["dil"], num_deformable_group=kw["dgroup"])
tx, to, tw = [torch.from_numpy(v).cuda() for v in (x, off, w)]
y = ops.deform_conv_forward(tx, to, tw, **a)
yu, _ = ops.deform_conv_forward(tx, to, tw, keep_col=True, **a)
e = (y - yu).abs()
print("max err", float(e.max()), "of", float(yu.abs().max()))
print("per image", e.amax(dim=(1, 2, 3)).tolist())
print("per filter (first 8)", e.amax(dim=(0, 2, 3))[:8].tolist())
ep = e.amax(dim=(0, 1)).reshape(-1)
print("bad pixels", (ep > 1e-3).nonzero().reshape(-1).tolist()[:40], "of", ep.numel())
# which groups are wrong: zero all weights except one group's channels
for grp in range(4):
    w2 = torch.zeros_like(tw); w2[:, grp * 16:(grp + 1) * 16] = tw[:, grp * 16:(grp + 1) * 16]
    y2 = ops.deform_conv_forward(tx, to, w2, **a); y2u, _ = ops.deform_conv_forward(tx, to, w2, keep_col=True, **a)
    print("group", grp, "max err", float((y2 - y2u).abs().max()))
    for tap in range(9):
        w3 = torch.zeros_like(tw); w3[:, grp * 16:(grp + 1) * 16, tap // 3, tap % 3] = tw[:, grp * 16:(grp + 1) * 16, tap // 3, tap % 3]
        y3 = ops.deform_conv_forward(tx, to, w3, **a); y3u, _ = ops.deform_conv_forward(tx, to, w3, keep_col=True, **a)
        er = float((y3 - y3u).abs().max())
        if er > 1e-4: print("   tap", tap, "err", er)
