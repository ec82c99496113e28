"""Pin the CPU oracle (oracle/*.c) against the reference's own outputs -- CPU only, no GPU.

Golden fixtures under tests/golden/ were produced by running the reference's own Python/Cython code
(tests/golden/make_golden.py); the ROIPooling vector is the reference's docstring example
(operator_cxx/roi_pooling_v1.cc:265-285); libc rand()/std::random_shuffle are called for real.
"""
import os

import numpy as np
import pytest

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def gold(name):
    return np.load(os.path.join(GOLD, name))


# ---------------------------------------------------------------------------------- anchors ----
@pytest.mark.parametrize("stride", [4, 8, 16, 32, 64])
def test_gen_anchor_matches_reference_numpy_twin(oracle, stride):
    g = gold("anchors.npz")["fpn_anchor_stride%d" % stride]
    n = g.shape[2]
    got = oracle.gen_anchor(n, n, stride, [8], [0.5, 1.0, 2.0])
    np.testing.assert_array_equal(got.reshape(g.shape), g)


def test_gen_anchor_c4_scales(oracle):
    g = gold("anchors.npz")["c4_anchor_stride16"]
    n = g.shape[2]
    got = oracle.gen_anchor(n, n, 16, [2, 4, 8, 16, 32], [0.5, 1.0, 2.0])
    np.testing.assert_array_equal(got.reshape(g.shape), g)


def test_base_anchor_known_answer(oracle):
    # SURVEY A.6 known answer for stride 4, ratio-major
    b = oracle.gen_base_anchors(4, [8], [0.5, 1, 2])
    np.testing.assert_array_equal(b, [[-22, -10, 25, 13], [-14, -14, 17, 17], [-10, -22, 13, 25]])


# ------------------------------------------------------------------------------- fpn assign ----
def test_fpn_assign_matches_reference_customop(oracle):
    g = gold("fpn_assign.npz")
    per, level = oracle.fpn_roi_assign(g["rois"], list(g["strides"]))
    np.testing.assert_array_equal(per, g["per_level"])
    # every roi lands on exactly one level
    assert np.all((level >= 0) & (level < 4))


# ---------------------------------------------------------------------------------- roipool ----
def test_roi_pool_reference_docstring_golden(oracle):
    x = np.arange(48, dtype=np.float32).reshape(1, 1, 8, 6)
    y = np.array([[0, 0, 0, 4, 4]], np.float32)
    out, idx = oracle.roi_pool_v1_fwd(x, y, (2, 2), 1.0)
    np.testing.assert_array_equal(out, [[[[14, 16], [26, 28]]]])
    np.testing.assert_array_equal(idx, [[[[14, 16], [26, 28]]]])  # values == flat indices here
    out, _ = oracle.roi_pool_v1_fwd(x, y, (2, 2), 0.7)
    np.testing.assert_array_equal(out, [[[[7, 9], [19, 21]]]])


def test_roi_pool_scatter_equals_cpu_gather(oracle):
    rs = np.random.RandomState(0)
    data = rs.standard_normal((2, 3, 12, 15)).astype(np.float32)
    rois = np.array([[0, 3, 2, 90, 70], [1, 0, 0, 239, 191], [0, 50, 60, 51, 61],
                     [1, 300, 300, 400, 400], [0, -20, -20, 30, 40]], np.float32)
    out, idx = oracle.roi_pool_v1_fwd(data, rois, (7, 7), 1 / 16.0)
    dy = rs.standard_normal(out.shape).astype(np.float32)
    a = oracle.roi_pool_v1_bwd(dy, rois, idx, data.shape, 1 / 16.0)
    b = oracle.roi_pool_v1_bwd(dy, rois, idx, data.shape, 1 / 16.0, gather=True)
    np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-6)
    # empty bins: value 0, argmax -1 (roi fully outside the map)
    assert np.all(out[3] == 0) and np.all(idx[3] == -1)


# ------------------------------------------------------------------------------ NMS family ----
@pytest.mark.parametrize("seed", [0, 1, 2])
def test_soft_nms_matches_reference_cython(oracle, seed):
    g = gold("cython_nms.npz")
    d = g["dets%d" % seed]
    for m, name in ((0, "hard"), (1, "linear"), (2, "gaussian")):
        b, i = oracle.soft_nms(d, 0.5, 0.3, 0.05, m)
        np.testing.assert_array_equal(b, g["soft_%s_boxes%d" % (name, seed)])
        np.testing.assert_array_equal(i, g["soft_%s_inds%d" % (name, seed)])


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_greedy_nms_and_overlaps_match_reference_cython(oracle, seed):
    g = gold("cython_nms.npz")
    d = g["dets%d" % seed]
    np.testing.assert_array_equal(oracle.greedy_nms(d, 0.45), g["greedy%d" % seed])
    np.testing.assert_array_equal(oracle.bbox_overlaps(d[:, :4], d[:40, :4]), g["overlaps%d" % seed])


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_contrib_nms_matches_reference_numpy_nms(oracle, seed):
    """_contrib_NMS (GPU path, suppress IoU > thr) and operator_py/nms.py:nms (keep IoU <= thr)
    use the same +1 IoU and the same threshold sense: same kept boxes in the same order."""
    g = gold("py_nms.npz")
    d = g["dets%d" % seed]
    kept = g["kept%d" % seed]
    out, score, keep = oracle.nms(d[None], -1, d.shape[0], 0.5)
    n = kept.shape[0]
    np.testing.assert_array_equal(out[0, :n], kept[:, :4])
    np.testing.assert_array_equal(score[0, :n, 0], kept[:, 4])
    assert np.all(out[0, n:] == 0) and np.all(keep[0, n:] == -1)


def test_live_reference_cython_if_built(oracle):
    """When oracle/_ref is present (built from /root/reference by oracle/build_ref.py) compare
    live on fresh random inputs, including exact score ties."""
    ref = pytest.importorskip("oracle._ref.cpu_nms")
    rs = np.random.RandomState(5)
    n = 200
    ctr = rs.rand(n, 2) * 200
    wh = rs.rand(n, 2) * 60 + 5
    sc = np.round(rs.rand(n, 1) * 8) / 8  # many exact ties
    d = np.concatenate([ctr - wh / 2, ctr + wh / 2, sc], 1).astype(np.float32)
    for m in (0, 1, 2):
        b, i = ref.soft_nms(d, np.float32(0.5), np.float32(0.3), np.float32(0.01), np.uint8(m))
        ob, oi = oracle.soft_nms(d, 0.5, 0.3, 0.01, m)
        np.testing.assert_array_equal(ob, b)
        np.testing.assert_array_equal(oi, np.asarray(i))


# ----------------------------------------------------------------------------- RNG / shuffle ----
def test_glibc_rand_restatement_matches_libc(oracle):
    for seed in (1, 0, 42, 2 ** 31 - 1):
        oracle.libc_srand(seed)
        want = [oracle.libc_rand() for _ in range(2000)]
        g = oracle.GlibcRand(seed)
        got = [g.next() for _ in range(2000)]
        assert got == want


def test_shuffle_restatement_matches_std_random_shuffle(oracle):
    import ctypes
    for n in (1, 2, 7, 100, 2100):
        oracle.libc_srand(7)
        want = oracle.std_random_shuffle(np.arange(n))
        # restated shuffle driven by the restated generator from the same seed
        g = oracle.GlibcRand(7)
        a = list(range(n))
        for i in range(1, n):
            j = g.next() % (i + 1)
            a[i], a[j] = a[j], a[i]
        np.testing.assert_array_equal(want, a)


def _pt_case(seed, B=2, N=300, M=20, n_gt=(5, 9)):
    from simpledet_amd import synth
    return synth.proposal_target_inputs(seed, B, N, M, n_gt)


def test_proposal_target_restated_rng_equals_libc_path(oracle):
    rois, gt = _pt_case(0)
    p = oracle.make_pt_param(81, 2, 128)
    oracle.libc_srand(1)
    a = oracle.proposal_target(rois, gt, p, use_libc=True)
    b = oracle.proposal_target(rois, gt, p, rng=oracle.GlibcRand(1))
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)
    ro, lb, bt, bw, iou, kept, rc = b
    assert rc == 0
    # structure: fg first with labels > 0 and iou >= 0.5, everything else label 0 and iou < 0.5
    for i in range(2):
        nfg = int((lb[i] > 0).sum())
        assert nfg <= 32 and np.all(lb[i, :nfg] > 0) and np.all(lb[i, nfg:] == 0)
        assert np.all(iou[i, :nfg] >= 0.5) and np.all(iou[i, nfg:] < 0.5)
        # 4-of-4K expansion: exactly the label's slot is weighted
        w = bw[i].reshape(128, 81, 4)
        assert np.all(w[np.arange(nfg), lb[i, :nfg].astype(int)] == 1)
        assert w.sum() == 4 * nfg
