"""ORACLE (test infrastructure, not product code) -- numpy restatement of the RPN anchor-target
assignment the reference runs in its data loader.

Follows
  core/detection_input.py   AnchorTarget2D: base_anchor :373-399, v/h_all_anchor :401-437,
                            _assign_label_to_anchor :450-482, _sample_anchor :484-499,
                            _cal_anchor_target :501-510, _gather_valid_anchor :512-520,
                            _scatter_valid_anchor :522-533, apply :535-565
  models/FPN/input.py       PyramidAnchorTarget2DBase.apply :19-50, PyramidAnchorTarget2D :53-146
  operator_py/bbox_transform.py  nonlinear_transform :52-77 (float64 arithmetic)
  operator_py/cython/bbox.pyx    bbox_overlaps_cython :31-72 (through oracle/liboracle.so, which is
                                 pinned bit for bit to the reference's compiled Cython)
  numpy  RandomState.choice(a, size, replace=False) == a[permutation(len(a))[:size]] with the
         legacy Fisher-Yates shuffle over MT19937 (restated in legacy_* below and pinned to numpy
         itself in tests/test_rpn_target.py)

Pinned: tests/golden/rpn_target.npz holds the outputs of the reference's own classes on the seeded
cases of tests/rpncases.py (tests/golden/make_golden_rpn.py), including the generator state after
the calls.  May be imported by tests/ only.
"""
import numpy as np

from . import pyoracle


# ------------------------------------------------------------------------------------------ anchors
def base_anchor(stride, scales, aspects):
    scales = np.atleast_1d(np.asarray(scales))
    aspects = np.atleast_1d(np.asarray(aspects))
    b = np.array([0, 0, stride - 1, stride - 1])
    w = b[2] - b[0] + 1
    h = b[3] - b[1] + 1
    x_ctr = b[0] + 0.5 * (w - 1)
    y_ctr = b[1] + 0.5 * (h - 1)
    w_ratios = np.round(np.sqrt(w * h / aspects))
    h_ratios = np.round(w_ratios * aspects)
    ws = np.outer(w_ratios, scales).reshape(-1)
    hs = np.outer(h_ratios, scales).reshape(-1)
    return np.stack([x_ctr - 0.5 * (ws - 1), y_ctr - 0.5 * (hs - 1), x_ctr + 0.5 * (ws - 1),
                     y_ctr + 0.5 * (hs - 1)], axis=1)


def level_anchors(stride, fh, fw, scales, aspects):
    """all anchors of one level, (fh*fw*A, 4) float64, index (y*fw + x)*A + a"""
    shift_x = np.arange(0, fw, dtype=np.float32) * stride
    shift_y = np.arange(0, fh, dtype=np.float32) * stride
    gx, gy = np.meshgrid(shift_x, shift_y)
    gx, gy = gx.reshape(-1), gy.reshape(-1)
    grid = np.stack([gx, gy, gx, gy], axis=1)
    return (grid[:, None, :] + base_anchor(stride, scales, aspects)[None, :, :]).reshape(-1, 4)


def level_shapes(cfg, portrait):
    """[(stride, fh, fw)] per level; h >= w ("v") uses (long, short)"""
    strides = cfg["stride"] if isinstance(cfg["stride"], (tuple, list)) else (cfg["stride"],)
    shorts = cfg["short"] if isinstance(cfg["short"], (tuple, list)) else (cfg["short"],)
    longs = cfg["long"] if isinstance(cfg["long"], (tuple, list)) else (cfg["long"],)
    return [(s, lg, sh) if portrait else (s, sh, lg) for s, sh, lg in zip(strides, shorts, longs)]


def all_anchors(cfg, portrait):
    return np.concatenate([level_anchors(s, fh, fw, cfg["scales"], cfg["aspects"])
                           for s, fh, fw in level_shapes(cfg, portrait)])


# ------------------------------------------------------------------------------ legacy numpy RNG --
class MT19937:
    """numpy's legacy generator: 624-word state + position; next_uint32 with tempering."""

    def __init__(self, key, pos):
        self.key = np.array(key, dtype=np.uint32).copy()
        self.pos = int(pos)
        self.draws = 0

    @classmethod
    def from_numpy(cls, rs):
        st = rs.get_state()
        return cls(st[1], st[2])

    def _twist(self):
        k = self.key.astype(np.uint64)
        for i in range(624):
            y = (k[i] & 0x80000000) | (k[(i + 1) % 624] & 0x7fffffff)
            k[i] = k[(i + 397) % 624] ^ (y >> np.uint64(1)) ^ (np.uint64(0x9908b0df) if (int(y) & 1) else np.uint64(0))
        self.key = k.astype(np.uint32)
        self.pos = 0

    def next_uint32(self):
        if self.pos >= 624:
            self._twist()
        y = int(self.key[self.pos])
        self.pos += 1
        self.draws += 1
        y ^= y >> 11
        y ^= (y << 7) & 0x9d2c5680
        y ^= (y << 15) & 0xefc60000
        y ^= y >> 18
        return y & 0xffffffff


def legacy_interval(mt, mx):
    """random_interval(): uniform integer in [0, mx] by masked rejection"""
    if mx == 0:
        return 0
    mask = mx
    for s in (1, 2, 4, 8, 16):
        mask |= mask >> s
    while True:
        v = mt.next_uint32() & mask
        if v <= mx:
            return v


def legacy_permutation(mt, n):
    """RandomState.permutation(n): arange(n) shuffled from the top (legacy _shuffle_raw)"""
    x = list(range(n))
    for i in range(n - 1, 0, -1):
        j = legacy_interval(mt, i)
        x[i], x[j] = x[j], x[i]
    return x


# ----------------------------------------------------------------------------------------- the op
def assign(valid_anchor, gt, neg_thr, pos_thr, min_pos_thr):
    n = valid_anchor.shape[0]
    label = np.full((n,), -1, np.float32)
    if len(gt) > 0:
        ov = pyoracle.bbox_overlaps(valid_anchor.astype(np.float32, copy=False), gt.astype(np.float32, copy=False))
        max_ov = ov.max(axis=1)
        argmax = ov.argmax(axis=1)
        gt_max = ov.max(axis=0)
        # the reference's (acknowledged) looseness: ANY anchor whose overlap equals a gt's maximum
        # and is >= min_pos_thr, including a maximum of 0 with min_pos_thr = 0
        gt_arg = np.where((ov == gt_max) & (ov >= min_pos_thr))[0]
        label[max_ov < neg_thr] = 0
        label[gt_arg] = 1
        label[max_ov >= pos_thr] = 1
    else:
        label[:] = 0
        argmax = np.zeros((n,))
    return label, argmax


def sample(label, num, fg_fraction, rs):
    num_fg = int(fg_fraction * num)
    fg = np.where(label == 1)[0]
    if len(fg) > num_fg:
        label[rs.choice(fg, size=(len(fg) - num_fg), replace=False)] = -1
    num_bg = num - np.sum(label == 1)
    bg = np.where(label == 0)[0]
    if len(bg) > num_bg:
        label[rs.choice(bg, size=(len(bg) - num_bg), replace=False)] = -1


def nonlinear_transform(ex, gt):
    ew = ex[:, 2] - ex[:, 0] + 1.0
    eh = ex[:, 3] - ex[:, 1] + 1.0
    ex_x = ex[:, 0] + 0.5 * (ew - 1.0)
    ex_y = ex[:, 1] + 0.5 * (eh - 1.0)
    gw = gt[:, 2] - gt[:, 0] + 1.0
    gh = gt[:, 3] - gt[:, 1] + 1.0
    gx = gt[:, 0] + 0.5 * (gw - 1.0)
    gy = gt[:, 1] + 0.5 * (gh - 1.0)
    return np.vstack(((gx - ex_x) / (ew + 1e-14), (gy - ex_y) / (eh + 1e-14), np.log(gw / ew),
                      np.log(gh / eh))).transpose()


def rpn_target_flat(im_info, gt_bbox, cfg, rs):
    """PyramidAnchorTarget2DBase.apply / the body of AnchorTarget2D.apply: flat all-anchor order.
    Returns (cls_label (N,), reg_target (N,4), reg_weight (N,4), portrait)."""
    gt = gt_bbox[np.where(gt_bbox[:, 0] != -1)[0]]
    if gt.shape[1] == 5:
        gt = gt[:, :4]
    h, w = im_info[:2]
    portrait = bool(h >= w)
    anchors = all_anchors(cfg, portrait)
    ab = cfg["allowed_border"]
    valid = np.where((anchors[:, 0] >= -ab) & (anchors[:, 1] >= -ab) & (anchors[:, 2] < w + ab) &
                     (anchors[:, 3] < h + ab))[0]
    va = anchors[valid]
    label, argmax = assign(va, gt, cfg["neg_thr"], cfg["pos_thr"], cfg["min_pos_thr"])
    sample(label, cfg["image_anchor"], cfg["pos_fraction"], rs)
    tgt = np.zeros((len(va), 4), np.float32)
    wgt = np.zeros((len(va), 4), np.float32)
    fg = np.where(label == 1)[0]
    if len(fg) > 0:
        tgt[fg] = nonlinear_transform(va[fg], gt[argmax[fg].astype(np.int64), :4])
        wgt[fg, :] = 1.0
    n = anchors.shape[0]
    cls = np.full((n,), -1, np.float32)
    rt = np.zeros((n, 4), np.float32)
    rw = np.zeros((n, 4), np.float32)
    cls[valid], rt[valid], rw[valid] = label, tgt, wgt
    return cls, rt, rw, portrait


def final_layout(cls, rt, rw, cfg, portrait):
    """AnchorTarget2D.apply :556-560 (single level) / PyramidAnchorTarget2D.apply :101-146"""
    pyramid = isinstance(cfg["stride"], (tuple, list))
    if not pyramid:
        (_, fh, fw), = level_shapes(cfg, portrait)
        return (cls.reshape((fh, fw, -1)).transpose(2, 0, 1).reshape(-1),
                rt.reshape((fh, fw, -1)).transpose(2, 0, 1), rw.reshape((fh, fw, -1)).transpose(2, 0, 1))
    cl, tl, wl = [], [], []
    off = 0
    na = len(np.atleast_1d(cfg["scales"])) * len(np.atleast_1d(cfg["aspects"]))
    for _, fh, fw in level_shapes(cfg, portrait):
        n = fh * fw * na
        cl.append(cls[off:off + n].reshape((fh, fw, -1)).transpose(2, 0, 1).reshape(-1, fh * fw))
        tl.append(rt[off:off + n].reshape((fh, fw, -1)).transpose(2, 0, 1).reshape(-1, fh * fw))
        wl.append(rw[off:off + n].reshape((fh, fw, -1)).transpose(2, 0, 1).reshape(-1, fh * fw))
        off += n
    return np.concatenate(cl, axis=1).reshape(-1), np.concatenate(tl, axis=1), np.concatenate(wl, axis=1)


def rpn_target(im_info, gt_bbox, cfg, rs):
    cls, rt, rw, portrait = rpn_target_flat(im_info, gt_bbox, cfg, rs)
    return final_layout(cls, rt, rw, cfg, portrait)
