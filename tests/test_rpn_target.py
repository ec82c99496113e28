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
