"""RPN anchor-target assignment: core/detection_input.py:345-565,
models/FPN/input.py:9-146.

tests/golden/rpn_target.npz was produced by the reference's OWN classes (loaded from the reference
files, real numpy RNG, the reference's compiled Cython IoU) on the seeded cases of tests/rpncases.py.
  CPU: the numpy oracle reproduces the fixtures bit for bit, including the MT19937 state after the
       calls; the restated legacy generator / shuffle equals numpy's.
  GPU: the HIP op reproduces labels, targets and the generator state bit for bit.
"""
import hashlib
import os

import numpy as np
import pytest

from . import rpncases

GOLD = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "rpn_target.npz"))


def _dense(name, i):
    shape = tuple(GOLD["%s/%d/shape" % (name, i)])
    tgt = np.zeros(int(np.prod(shape)), np.float32)
    wgt = np.zeros(int(np.prod(shape)), np.float32)
    nz = GOLD["%s/%d/nz" % (name, i)]
    tgt[nz] = GOLD["%s/%d/target_nz" % (name, i)]
    wgt[nz] = 1.0
    return GOLD["%s/%d/label" % (name, i)].astype(np.float32), tgt.reshape(shape), wgt.reshape(shape)


def test_restated_legacy_rng_equals_numpy():
    from oracle import rpn_target as orc
    for seed, n in [(0, 1), (1, 2), (2, 17), (3, 1000), (4, 4097)]:
        rs = np.random.RandomState(seed)
        mt = orc.MT19937.from_numpy(rs)
        want = rs.permutation(n)
        got = orc.legacy_permutation(mt, n)
        assert got == want.tolist()
        st = rs.get_state()
        # same number of outputs consumed: positions agree modulo the 624-word block
        assert mt.pos == st[2] and np.array_equal(mt.key, st[1])
    # choice(replace=False) is a[permutation(n)[:size]]
    rs, rs2 = np.random.RandomState(9), np.random.RandomState(9)
    a = np.arange(100, 300)
    assert np.array_equal(rs.choice(a, size=37, replace=False), a[rs2.permutation(200)[:37]])


@pytest.mark.parametrize("name", sorted(rpncases.CASES))
def test_anchors_equal_reference(name):
    from oracle import rpn_target as orc
    cfg = rpncases.CASES[name]["cfg"]
    for orient, portrait in (("h", False), ("v", True)):
        a = np.ascontiguousarray(orc.all_anchors(cfg, portrait), np.float64)
        want = bytes(GOLD["%s/%s_all_anchor_sha256" % (name, orient)])
        assert hashlib.sha256(a.tobytes()).digest() == want
        # the device op takes them as float32: they are exactly representable
        assert np.array_equal(a.astype(np.float32).astype(np.float64), a)


@pytest.mark.parametrize("name", sorted(rpncases.CASES))
def test_oracle_reproduces_reference(name):
    from oracle import rpn_target as orc
    case = rpncases.CASES[name]
    rs = np.random.RandomState(case["seed"])
    for i, (im_info, gt) in enumerate(rpncases.inputs(case)):
        lab, tgt, wgt = orc.rpn_target(im_info, gt, case["cfg"], rs)
        wl, wt, ww = _dense(name, i)
        np.testing.assert_array_equal(lab, wl)
        np.testing.assert_array_equal(np.asarray(tgt, np.float32), wt)
        np.testing.assert_array_equal(np.asarray(wgt, np.float32), ww)
    st = rs.get_state()
    assert np.array_equal(st[1], GOLD[name + "/mt_key"]) and st[2] == int(GOLD[name + "/mt_pos"][0])


def test_zero_overlap_gt_quirk_is_in_the_fixture():
    """min_pos_thr = 0 and a gt box nothing overlaps: the reference marks every valid anchor
    positive before sampling (its own TODO); then exactly image_anchor * pos_fraction survive and no
    background is left."""
    lab, _, _ = _dense("fpn_zero_overlap_gt", 0)
    assert (lab == 1).sum() == 128 and (lab == 0).sum() == 0


# ------------------------------------------------------------------------------------------ GPU --
@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(rpncases.CASES))
def test_hip_reproduces_reference(ops, name):
    """labels, box targets, weights AND the MT19937 state after the calls, bit for bit against what
    the reference's own classes produced (images of a case go through one generator state)."""
    import torch
    case = rpncases.CASES[name]
    cfg = case["cfg"]
    p = ops.rpn_target_param(cfg["stride"], cfg["short"], cfg["long"], cfg["scales"], cfg["aspects"],
                             cfg["allowed_border"], cfg["pos_thr"], cfg["neg_thr"], cfg["min_pos_thr"],
                             cfg["image_anchor"], cfg["pos_fraction"])
    state = ops.mt19937_state(seed=case["seed"])
    ref_state = np.random.RandomState(case["seed"]).get_state()
    np.testing.assert_array_equal(state.cpu().numpy()[:624].astype(np.uint32), ref_state[1])
    ins = rpncases.inputs(case)
    for i, (im_info, gt) in enumerate(ins):  # one image per call: orientations may differ
        cls, tgt, wgt = ops.rpn_anchor_target(torch.from_numpy(im_info[None]).cuda(),
                                              torch.from_numpy(gt[None]).cuda(), p, state, layout=1)
        wl, wt, ww = _dense(name, i)
        np.testing.assert_array_equal(cls[0].cpu().numpy(), wl)
        np.testing.assert_array_equal(wgt[0].cpu().numpy().reshape(ww.shape), ww)
        np.testing.assert_array_equal(tgt[0].cpu().numpy().reshape(wt.shape), wt)
    st = state.cpu().numpy()
    np.testing.assert_array_equal(st[:624].astype(np.uint32), GOLD[name + "/mt_key"])
    assert int(st[624]) == int(GOLD[name + "/mt_pos"][0])


@pytest.mark.gpu
def test_hip_batch_equals_image_by_image_and_flat_layout(ops):
    """B = 2 in one call == two calls (the state carries), and layout 0 is the flat all-anchor order
    of PyramidAnchorTarget2DBase."""
    import torch
    from oracle import rpn_target as orc
    case = rpncases.CASES["fpn_landscape"]
    cfg = case["cfg"]
    p = ops.rpn_target_param(cfg["stride"], cfg["short"], cfg["long"], cfg["scales"], cfg["aspects"])
    ins = rpncases.inputs(case)
    im = torch.from_numpy(np.stack([x[0] for x in ins])).cuda()
    gt = torch.from_numpy(np.stack([x[1] for x in ins])).cuda()
    state = ops.mt19937_state(seed=case["seed"])
    cls, tgt, wgt = ops.rpn_anchor_target(im, gt, p, state, layout=0)
    rs = np.random.RandomState(case["seed"])
    for i, (im_info, g) in enumerate(ins):
        c, t, w, _ = orc.rpn_target_flat(im_info, g, cfg, rs)
        np.testing.assert_array_equal(cls[i].cpu().numpy(), c)
        np.testing.assert_array_equal(tgt[i].cpu().numpy(), t)
        np.testing.assert_array_equal(wgt[i].cpu().numpy(), w)
    assert int(state[624]) == rs.get_state()[2]


@pytest.mark.gpu
def test_hip_generator_replay_fuzz(ops):
    """160 small random problems through ONE generator state against numpy's RandomState (the oracle
    restatement, itself pinned to the reference's classes): list lengths from 3 to ~3000 cross every
    rejection-mask boundary many times, at arbitrary positions of the 64-draw batches and of the
    624-word state, with fg-only, bg-only and both subsamplings.  Labels and the state position must
    agree after every image."""
    import torch
    from oracle import rpn_target as orc
    rs_cfg = np.random.RandomState(123)
    state = ops.mt19937_state(seed=77)
    rs = np.random.RandomState(77)
    for it in range(160):
        short, long_ = int(rs_cfg.randint(3, 26)), int(rs_cfg.randint(26, 40))
        cfg = dict(stride=16, short=short, long=long_, scales=(2, 4, 8)[:int(rs_cfg.randint(1, 4))],
                   aspects=(0.5, 1.0, 2.0), allowed_border=int(rs_cfg.choice([0, 16, 64])),
                   pos_thr=float(rs_cfg.choice([0.5, 0.7])), neg_thr=0.3, min_pos_thr=0.0,
                   image_anchor=int(rs_cfg.choice([4, 16, 64, 256])), pos_fraction=float(rs_cfg.choice([0.25, 0.5])))
        h, w = short * 16, long_ * 16
        ng = int(rs_cfg.randint(0, 12))
        gt = -np.ones((100, 5), np.float32)
        if ng:
            x1 = rs_cfg.rand(ng) * (w - 40)
            y1 = rs_cfg.rand(ng) * (h - 40)
            bw = 16 + rs_cfg.rand(ng) * (w - x1 - 17)
            bh = 16 + rs_cfg.rand(ng) * (h - y1 - 17)
            gt[:ng] = np.stack([x1, y1, x1 + bw, y1 + bh, np.ones(ng)], 1).astype(np.float32)
        im_info = np.array([h, w, 1.0], np.float32)
        p = ops.rpn_target_param(cfg["stride"], cfg["short"], cfg["long"], cfg["scales"], cfg["aspects"],
                                 cfg["allowed_border"], cfg["pos_thr"], cfg["neg_thr"], cfg["min_pos_thr"],
                                 cfg["image_anchor"], cfg["pos_fraction"])
        cls, tgt, wgt = ops.rpn_anchor_target(torch.from_numpy(im_info[None]).cuda(),
                                              torch.from_numpy(gt[None]).cuda(), p, state, layout=0)
        c, t, wv, _ = orc.rpn_target_flat(im_info, gt, cfg, rs)
        np.testing.assert_array_equal(cls[0].cpu().numpy(), c, err_msg="labels, problem %d" % it)
        np.testing.assert_array_equal(wgt[0].cpu().numpy(), wv, err_msg="weights, problem %d" % it)
        np.testing.assert_array_equal(tgt[0].cpu().numpy(), t, err_msg="targets, problem %d" % it)
        assert int(state[624]) == rs.get_state()[2], "generator position, problem %d" % it
    np.testing.assert_array_equal(state.cpu().numpy()[:624].astype(np.uint32), rs.get_state()[1])


def _batched_replay(mt, n, keep, batch=64):
    """The device's consumption rule (csrc/rpn_target.hip rpn_sample) in plain Python: `batch` outputs
    at a time, exact acceptance inside the batch, and a batch is CUT at the accept that ends a
    rejection-mask segment -- the draws behind it are read again with the next (halved) mask."""
    def temper(y):
        y ^= y >> 11
        y ^= (y << 7) & 0x9d2c5680
        y ^= (y << 15) & 0xefc60000
        y ^= y >> 18
        return y & 0xffffffff
    i, jrec = n - 1, {}
    while i >= 1:
        mask = i
        for s in (1, 2, 4, 8, 16):
            mask |= mask >> s
        lo = (mask >> 1) + 1
        if mt.pos >= 624:
            mt._twist()
        chunk = min(624 - mt.pos, batch)
        v = [temper(int(mt.key[mt.pos + t])) & mask for t in range(chunk)]
        acc, c = [], 0
        for t in range(chunk):
            a = v[t] <= i - c
            acc.append(a)
            c += a
        seg = i - lo + 1
        consumed = chunk
        if c >= seg:
            cnt = 0
            for t in range(chunk):
                cnt += acc[t]
                if cnt == seg:
                    consumed = t + 1
                    break
            acc = [a and t < consumed for t, a in enumerate(acc)]
            c = seg
        cnt = 0
        for t in range(consumed):
            if acc[t]:
                if (n - 1) - (i - cnt) < keep:
                    jrec[(n - 1) - (i - cnt)] = v[t]
                cnt += 1
        i -= c
        mt.pos += consumed
    return jrec


def test_batched_rejection_replay_equals_serial_numpy_walk():
    """400 permutations of random length through one generator: the batched rule consumes exactly
    numpy's draws (same swap targets for the kept prefix, same state).  Cutting only when MORE
    accepts follow the segment's last one (the first version of the kernel) loses ~2 % of them."""
    from oracle import rpn_target as orc
    st = np.random.RandomState(9).get_state()
    a, b = orc.MT19937(st[1], st[2]), orc.MT19937(st[1], st[2])
    rs = np.random.RandomState(5)
    for _ in range(400):
        n, keep = int(rs.randint(3, 3000)), int(rs.randint(1, 64))
        ja = _batched_replay(a, n, keep)
        jb = {s: orc.legacy_interval(b, i) for s, i in enumerate(range(n - 1, 0, -1))}
        assert ja == {s: j for s, j in jb.items() if s < keep}
        assert a.pos == b.pos and (a.key == b.key).all()
