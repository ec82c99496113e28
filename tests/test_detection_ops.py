"""HIP-vs-oracle parity for ROIPooling_v1, GenAnchor, _contrib_NMS, soft-NMS, bbox_overlaps and
ProposalTarget (GPU), through the C ABI.  Bars: everything integer / index / selection is bit-exact;
floats produced by +,-,*,/ are bit-exact (IEEE ops in the reference's order); the only tolerance
is 1e-6 relative on the two log() box deltas of ProposalTarget (device logf vs glibc logf)."""
import os

import numpy as np
import pytest

from simpledet_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# ------------------------------------------------------------------------------------- RoIPool --
def _pool_case(seed, B=2, C=5, H=25, W=42, K=40, stride=32):
    rs = np.random.RandomState(seed)
    data = rs.standard_normal((B, C, H, W)).astype(np.float32)
    r = synth.random_rois(seed, 1, K, H * stride, W * stride)[0]
    rois = np.concatenate([rs.randint(0, B, (K, 1)).astype(np.float32), r], 1)
    return data, rois


@pytest.mark.gpu
def test_roi_pool_reference_docstring_golden_on_gpu(ops):
    x = np.arange(48, dtype=np.float32).reshape(1, 1, 8, 6)
    y = np.array([[0, 0, 0, 4, 4]], np.float32)
    out, idx = ops.roi_pool_v1_forward(_t(x), _t(y), (2, 2), 1.0)
    np.testing.assert_array_equal(out.cpu().numpy(), [[[[14, 16], [26, 28]]]])
    out, _ = ops.roi_pool_v1_forward(_t(x), _t(y), (2, 2), 0.7)
    np.testing.assert_array_equal(out.cpu().numpy(), [[[[7, 9], [19, 21]]]])


@pytest.mark.gpu
@pytest.mark.parametrize("case", [(0, 7, 7, 1 / 32.0), (1, 7, 7, 1 / 16.0), (2, 3, 5, 1 / 32.0),
                                  (3, 14, 14, 1 / 32.0)])
@pytest.mark.parametrize("fwd", [1, 2, 0])  # four planes in LDS (default), one plane, wave per (roi, channel)
def test_roi_pool_forward_backward(ops, oracle, case, fwd):
    from simpledet_amd._lib import lib
    seed, ph, pw, scale = case
    data, rois = _pool_case(seed)
    want, widx = oracle.roi_pool_v1_fwd(data, rois, (ph, pw), scale)
    lib().set_tuning("roi_pool_fwd", fwd)
    try:
        out, idx = ops.roi_pool_v1_forward(_t(data), _t(rois), (ph, pw), scale)
    finally:
        lib().set_tuning("roi_pool_fwd", 1)
    np.testing.assert_array_equal(out.cpu().numpy(), want)
    np.testing.assert_array_equal(idx.cpu().numpy(), widx)
    dy = np.random.RandomState(9).standard_normal(want.shape).astype(np.float32)
    wdx = oracle.roi_pool_v1_bwd(dy, rois, widx, data.shape, scale)
    dx, drois = ops.roi_pool_v1_backward(_t(dy), _t(rois), idx, data.shape, scale)
    np.testing.assert_allclose(dx.cpu().numpy(), wdx, rtol=1e-5, atol=1e-5)
    assert float(drois.abs().max()) == 0
    # kAddTo accumulates
    dx2, _ = ops.roi_pool_v1_backward(_t(dy), _t(rois), idx, data.shape, scale, req_data="add",
                                      req_rois="null", d_data=dx.clone())
    np.testing.assert_allclose(dx2.cpu().numpy(), 2 * wdx, rtol=1e-5, atol=1e-5)
    with pytest.raises(RuntimeError, match="kWriteInplace"):
        ops.roi_pool_v1_backward(_t(dy), _t(rois), idx, data.shape, scale, req_data=2)


@pytest.mark.gpu
def test_roi_pool_forward_kernels_agree_at_c4_size(ops):
    """C4 baseline shape (1024 RoIs x 1024 channels on a 50x84 map, two images, a few batch indices
    out of range, more RoIs than one 512-RoI chunk): the plane-in-LDS kernel and the
    wave-per-(roi, channel) kernel give the same bits."""
    import torch
    from simpledet_amd._lib import lib
    g = torch.Generator(device="cuda").manual_seed(5)
    data = torch.randn((2, 1024, 50, 84), device="cuda", generator=g)
    r = synth.random_rois(5, 1, 1024)[0]
    bi = np.random.RandomState(5).randint(0, 2, (1024, 1)).astype(np.float32)
    bi[[7, 600]] = [[5.0], [-2.0]]
    rois = _t(np.concatenate([bi, r], 1))
    o1, i1 = ops.roi_pool_v1_forward(data, rois, (7, 7), 1 / 16.0)
    for mode in (0, 2):
        lib().set_tuning("roi_pool_fwd", mode)
        try:
            o0, i0 = ops.roi_pool_v1_forward(data, rois, (7, 7), 1 / 16.0)
        finally:
            lib().set_tuning("roi_pool_fwd", 1)
        assert torch.equal(o1, o0) and torch.equal(i1, i0), mode
    assert float(o1[[7, 600]].abs().max()) == 0 and float((i1[[7, 600]] + 1).abs().max()) == 0


@pytest.mark.gpu
def test_roi_pool_backward_bands_and_fallback(ops, oracle):
    """A plane larger than one LDS band (2 bands of a 120x100 map) and the global-atomic structure
    (knob roi_pool_bwd=0) against the LDS-plane kernel; conservation / arg-max properties at the C4
    baseline shape; a batch index outside [0, B) (undefined behaviour in the reference, which reads
    out of bounds) is rejected defensively: 0 / -1 forward, no gradient."""
    import torch
    from simpledet_amd._lib import lib
    data, rois = _pool_case(4, B=3, C=3, H=120, W=100, K=60, stride=8)
    scale = 1 / 8.0
    want, widx = oracle.roi_pool_v1_fwd(data, rois, (7, 7), scale)
    out, idx = ops.roi_pool_v1_forward(_t(data), _t(rois), (7, 7), scale)
    np.testing.assert_array_equal(out.cpu().numpy(), want)
    np.testing.assert_array_equal(idx.cpu().numpy(), widx)
    dy = np.random.RandomState(3).standard_normal(want.shape).astype(np.float32)
    wdx = oracle.roi_pool_v1_bwd(dy, rois, widx, data.shape, scale)
    dx, _ = ops.roi_pool_v1_backward(_t(dy), _t(rois), idx, data.shape, scale)
    np.testing.assert_allclose(dx.cpu().numpy(), wdx, rtol=1e-5, atol=1e-5)
    lib().set_tuning("roi_pool_bwd", 0)
    try:
        dx0, _ = ops.roi_pool_v1_backward(_t(dy), _t(rois), idx, data.shape, scale)
    finally:
        lib().set_tuning("roi_pool_bwd", 1)
    np.testing.assert_allclose(dx0.cpu().numpy(), wdx, rtol=1e-5, atol=1e-5)
    # conservation at the C4 baseline shape: every pooled gradient lands in exactly one pixel
    g = torch.Generator(device="cuda").manual_seed(0)
    big = torch.randn((2, 64, 50, 84), device="cuda", generator=g)
    r = synth.random_rois(1, 1, 256)[0]
    br = _t(np.concatenate([np.random.RandomState(1).randint(0, 2, (256, 1)).astype(np.float32), r], 1))
    o, ix = ops.roi_pool_v1_forward(big, br, (7, 7), 1 / 16.0)
    gy = torch.rand_like(o)
    gx, _ = ops.roi_pool_v1_backward(gy, br, ix, big.shape, 1 / 16.0)
    # (64 channels: the four-channel kernel) == the one-channel LDS kernel == the global-atomic structure,
    # also with kAddTo
    for mode in (2, 0):
        lib().set_tuning("roi_pool_bwd", mode)
        try:
            gm, _ = ops.roi_pool_v1_backward(gy, br, ix, big.shape, 1 / 16.0)
        finally:
            lib().set_tuning("roi_pool_bwd", 1)
        assert float((gm - gx).abs().max()) <= 1e-5 * float(gx.abs().max()), mode
    ga, _ = ops.roi_pool_v1_backward(gy, br, ix, big.shape, 1 / 16.0, req_data="add", req_rois="null",
                                     d_data=gx.clone())
    assert float((ga - 2 * gx).abs().max()) <= 1e-5 * float(gx.abs().max())
    tot = float((gy * (ix >= 0)).double().sum())
    assert abs(float(gx.double().sum()) - tot) <= 1e-6 * max(1.0, abs(tot))
    # the pooled value is the feature at the recorded arg-max
    flat = big.reshape(2, 64, -1)
    b = br[:, 0].long()
    sel = ix >= 0
    gathered = torch.gather(flat[b], 2, ix.clamp(min=0).long().reshape(256, 64, 49)).reshape(o.shape)
    assert torch.equal(gathered[sel], o[sel])
    bad = br.clone()
    bad[3, 0] = 7
    bad[4, 0] = -1
    o2, ix2 = ops.roi_pool_v1_forward(big, bad, (7, 7), 1 / 16.0)
    assert float(o2[3:5].abs().max()) == 0 and float((ix2[3:5] + 1).abs().max()) == 0
    gx2, _ = ops.roi_pool_v1_backward(gy, bad, ix2, big.shape, 1 / 16.0)
    keep = torch.ones(256, dtype=torch.bool, device="cuda")
    keep[3:5] = False
    tot2 = float((gy * (ix2 >= 0))[keep].double().sum())
    assert abs(float(gx2.double().sum()) - tot2) <= 1e-6 * max(1.0, abs(tot2))


# ----------------------------------------------------------------------------------- GenAnchor --
@pytest.mark.gpu
@pytest.mark.parametrize("stride,shape", [(4, (200, 334)), (8, (100, 167)), (16, (50, 84)),
                                          (32, (25, 42)), (64, (13, 21))])
def test_gen_anchor_fpn_levels_bit_exact(ops, oracle, stride, shape):
    h, w = shape
    got = ops.gen_anchor(h, w, stride, [8], [0.5, 1.0, 2.0]).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.gen_anchor(h, w, stride, [8], [0.5, 1.0, 2.0]))


@pytest.mark.gpu
def test_gen_anchor_all_levels_in_one_launch(ops, oracle):
    """sd_gen_anchor_levels == one sd_gen_anchor per level, bit for bit (P2-P6, the reference's
    scales / ratios), including an empty level and more anchors per location than the one-launch
    kernel holds (falls back to a launch per level)."""
    shapes = [(200, 334), (100, 167), (50, 84), (25, 42), (13, 21), (0, 5)]
    strides = [4, 8, 16, 32, 64, 128]
    for scales, ratios in (([8], [0.5, 1.0, 2.0]), ([2, 4, 8, 16, 32], [0.5, 1.0, 2.0])):
        got = ops.gen_anchor_levels(shapes, strides, scales, ratios)
        for g, (h, w), st in zip(got, shapes, strides):
            np.testing.assert_array_equal(g.cpu().numpy(), oracle.gen_anchor(h, w, st, scales, ratios))


@pytest.mark.gpu
def test_gen_anchor_matches_reference_numpy_twin_fixture(ops):
    g = np.load(os.path.join(GOLD, "anchors.npz"))
    for stride in (4, 8, 16, 32, 64):
        ref = g["fpn_anchor_stride%d" % stride]
        n = ref.shape[2]
        got = ops.gen_anchor(n, n, stride, [8], [0.5, 1.0, 2.0]).cpu().numpy()
        np.testing.assert_array_equal(got.reshape(ref.shape), ref)
    ref = g["c4_anchor_stride16"]
    n = ref.shape[2]
    got = ops.gen_anchor(n, n, 16, [2, 4, 8, 16, 32], [0.5, 1.0, 2.0]).cpu().numpy()
    np.testing.assert_array_equal(got.reshape(ref.shape), ref)


@pytest.mark.gpu
def test_gen_anchor_odd_ratios_and_errors(ops, oracle):
    got = ops.gen_anchor(7, 9, 16, [1.5, 3, 20], [0.3, 0.77, 4.2]).cpu().numpy()
    np.testing.assert_array_equal(got, oracle.gen_anchor(7, 9, 16, [1.5, 3, 20], [0.3, 0.77, 4.2]))
    assert ops.gen_anchor(0, 5, 16, [8], [1.0]).shape == (0, 4)
    with pytest.raises(RuntimeError):
        ops.gen_anchor(4, 4, 16, [8], [-1.0])


# ---------------------------------------------------------------------------------- hard NMS ----
def _nms_batch(seeds, n, mode="clustered"):
    return np.stack([synth.nms_dets(s, n, mode=mode) for s in seeds])


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [
    dict(n=2000, pre=-1, post=1000, thr=0.7), dict(n=2000, pre=1000, post=300, thr=0.7),
    dict(n=1000, pre=6000, post=1000, thr=0.5), dict(n=777, pre=500, post=600, thr=0.3),
    dict(n=64, pre=-1, post=64, thr=0.5), dict(n=65, pre=-1, post=10, thr=0.5),
    dict(n=1, pre=-1, post=1, thr=0.5), dict(n=5000, pre=-1, post=2000, thr=0.7)])
def test_nms_matches_oracle(ops, oracle, cfg):
    dets = _nms_batch([1, 2, 3], cfg["n"])
    want = oracle.nms(dets, cfg["pre"], cfg["post"], cfg["thr"])
    out, score, keep = ops.nms(_t(dets), cfg["pre"], cfg["post"], cfg["thr"], return_index=True)
    np.testing.assert_array_equal(keep.cpu().numpy(), want[2])
    np.testing.assert_array_equal(out.cpu().numpy(), want[0])
    np.testing.assert_array_equal(score.cpu().numpy(), want[1])


@pytest.mark.gpu
def test_nms_fuzz(ops, oracle):
    """100 random problems: 1..2500 boxes, batch 1..3, clustered / disjoint / piled-up boxes, score
    ties from none to eight distinct values, every pre / post / threshold combination (> and >=)."""
    rs = np.random.RandomState(31)
    for it in range(100):
        n = int(rs.choice([1, 2, 63, 64, 65, 127, 300, 1000, 2047, 2048, 2500]))
        B = int(rs.randint(1, 4))
        mode = str(rs.choice(["clustered", "no_overlap", "all_overlap"]))
        dets = _nms_batch([int(x) for x in rs.randint(0, 10000, B)], n, mode)
        q = int(rs.choice([0, 0, 8, 64]))
        if q:
            dets[:, :, 4] = np.round(dets[:, :, 4] * q) / q
        pre = int(rs.choice([-1, n, max(1, n // 2), 6000]))
        post = int(rs.choice([1, max(1, n // 3), n, 3000]))
        thr = float(rs.choice([0.3, 0.5, 0.7]))
        ge = bool(rs.randint(0, 2))
        want = oracle.nms(dets, pre, post, thr) if not ge else None
        out, score, keep = ops.nms(_t(dets), pre, post, thr, threshold_ge=ge, return_index=True)
        msg = "problem %d n=%d B=%d %s q=%d pre=%d post=%d thr=%g ge=%d" % (it, n, B, mode, q, pre, post, thr, ge)
        if want is not None:
            np.testing.assert_array_equal(keep.cpu().numpy(), want[2], err_msg=msg)
            np.testing.assert_array_equal(out.cpu().numpy(), want[0], err_msg=msg)
            np.testing.assert_array_equal(score.cpu().numpy(), want[1], err_msg=msg)
        else:  # >= variant: a kept box suppresses everything with IoU >= thr that follows it
            k = keep.cpu().numpy()
            for b in range(B):
                kb = k[b][k[b] >= 0]
                assert len(set(kb.tolist())) == len(kb), msg
                if len(kb) > 1 and n <= 300:
                    from oracle import pyoracle
                    bx = dets[b, kb, :4]
                    ov = pyoracle.bbox_overlaps(bx, bx)
                    assert (np.triu(ov, 1) < thr).all(), msg


@pytest.mark.gpu
def test_soft_nms_fuzz(ops, oracle):
    """60 random batches of problems (0..400 boxes each, all three methods, random sigma / Nt /
    score threshold, ties and duplicated boxes) against the oracle (pinned to the reference's
    compiled Cython)."""
    rs = np.random.RandomState(32)
    for it in range(60):
        P, Nmax = int(rs.randint(1, 7)), int(rs.choice([1, 17, 64, 200, 400]))
        dets = np.stack([synth.nms_dets(int(rs.randint(0, 10000)), Nmax,
                                        mode=str(rs.choice(["clustered", "no_overlap", "all_overlap"])))
                         for _ in range(P)])
        if rs.randint(0, 2):
            dets[:, :, 4] = np.round(dets[:, :, 4] * 16) / 16
        if Nmax >= 64 and rs.randint(0, 2):
            dets[:, Nmax // 2:Nmax // 2 + 20] = dets[:, :20]
        counts = rs.randint(0, Nmax + 1, P)
        _soft_check(ops, oracle, dets, counts, float(rs.choice([0.3, 0.5, 0.8])), float(rs.choice([0.3, 0.5])),
                    float(rs.choice([0.001, 0.05, 0.3])), int(rs.randint(0, 3)))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["no_overlap", "all_overlap"])
def test_nms_stress_sets_and_ties(ops, oracle, mode):
    dets = _nms_batch([4, 5], 1000, mode)
    dets[:, :, 4] = np.round(dets[:, :, 4] * 16) / 16  # many exact score ties: stable order matters
    want = oracle.nms(dets, -1, 1000, 0.5)
    out, score, keep = ops.nms(_t(dets), -1, 1000, 0.5, return_index=True)
    np.testing.assert_array_equal(keep.cpu().numpy(), want[2])
    np.testing.assert_array_equal(out.cpu().numpy(), want[0])


@pytest.mark.gpu
def test_nms_already_sorted_and_ge_threshold(ops, oracle):
    dets = _nms_batch([7], 600)
    order = np.argsort(-dets[0, :, 4], kind="stable")
    sd = dets[:, order]
    want = oracle.nms(sd, -1, 600, 0.6, already_sorted=True)
    out, score = ops.nms(_t(sd), -1, 600, 0.6, already_sorted=True)
    np.testing.assert_array_equal(out.cpu().numpy(), want[0])
    # >= variant (proposal_v3.cu:319 / greedy_nms): equals the reference's Cython greedy_nms set
    g = np.load(os.path.join(GOLD, "cython_nms.npz"))
    d = g["dets0"]
    _, _, keep = ops.nms(_t(d[None]), -1, d.shape[0], 0.45, threshold_ge=True, return_index=True)
    k = keep.cpu().numpy()[0]
    np.testing.assert_array_equal(np.sort(k[k >= 0]), g["greedy0"])


@pytest.mark.gpu
def test_nms_matches_reference_numpy_nms_fixture(ops):
    g = np.load(os.path.join(GOLD, "py_nms.npz"))
    for seed in range(3):
        d, kept = g["dets%d" % seed], g["kept%d" % seed]
        out, score = ops.nms(_t(d[None]), -1, d.shape[0], 0.5)
        n = kept.shape[0]
        np.testing.assert_array_equal(out.cpu().numpy()[0, :n], kept[:, :4])
        np.testing.assert_array_equal(score.cpu().numpy()[0, :n, 0], kept[:, 4])
        assert float(out[0, n:].abs().max()) == 0


@pytest.mark.gpu
def test_nms_workspace_and_size_errors(ops):
    import torch
    from simpledet_amd._lib import lib, SimpleDetOpsError
    import ctypes
    d = _t(_nms_batch([1], 100))
    out = torch.empty((1, 100, 4), device="cuda")
    sc = torch.empty((1, 100), device="cuda")
    with pytest.raises(SimpleDetOpsError, match="workspace"):
        lib().call("sd_nms", ctypes.c_void_p(d.data_ptr()), 1, 100, -1, 100, 0.5, 0, 0,
                   ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(sc.data_ptr()), None, None, 0,
                   None)
    big = torch.zeros((1, 20000, 5), device="cuda")
    with pytest.raises(SimpleDetOpsError, match="sort capacity"):
        ops.nms(big, -1, 10, 0.5)


# ---------------------------------------------------------------------------------- soft NMS ----
def _soft_expect(oracle, dets, counts, sigma, Nt, thr, method):
    res = []
    for p in range(dets.shape[0]):
        res.append(oracle.soft_nms(dets[p, :counts[p]], sigma, Nt, thr, method))
    return res


def _soft_check(ops, oracle, dets, counts, sigma, Nt, thr, method):
    import torch
    od, oi, oc = ops.soft_nms_batched(_t(dets), _t(counts.astype(np.int32)), sigma, Nt, thr, method)
    od, oi, oc = od.cpu().numpy(), oi.cpu().numpy(), oc.cpu().numpy()
    for p, (wb, wi) in enumerate(_soft_expect(oracle, dets, counts, sigma, Nt, thr, method)):
        assert oc[p] == len(wi), (p, oc[p], len(wi))
        np.testing.assert_array_equal(oi[p, :oc[p]], wi)
        np.testing.assert_array_equal(od[p, :oc[p]], wb)


@pytest.mark.gpu
@pytest.mark.parametrize("method", [0, 1, 2])
def test_soft_nms_matches_oracle(ops, oracle, method):
    rs = np.random.RandomState(method)
    P, Nmax = 12, 300
    dets = np.stack([synth.nms_dets(100 + p, Nmax) for p in range(P)])
    counts = np.array([300, 299, 257, 256, 255, 129, 64, 63, 2, 1, 0, 300])
    _soft_check(ops, oracle, dets, counts, 0.5, 0.3, 0.05, method)
    # reference defaults (detection_test.py: Nt=thresh 0.5, score_thresh 0.001)
    _soft_check(ops, oracle, dets, counts, 0.5, 0.5, 0.001, method)


@pytest.mark.gpu
def test_soft_nms_ties_duplicates_and_full_size(ops, oracle):
    # exact score ties + duplicated boxes: the selection order must follow position order
    d = synth.nms_dets(3, 400)
    d[:, 4] = np.round(d[:, 4] * 8) / 8
    d[100:150] = d[:50]
    dets = np.stack([d, synth.nms_dets(4, 400, mode="all_overlap"),
                     synth.nms_dets(5, 400, mode="no_overlap")])
    counts = np.array([400, 400, 400])
    for method in (0, 1, 2):
        _soft_check(ops, oracle, dets, counts, 0.5, 0.3, 0.01, method)
    # the baseline problem size: 1000 boxes
    big = np.stack([synth.nms_dets(50 + p, 1000) for p in range(4)])
    _soft_check(ops, oracle, big, np.array([1000] * 4), 0.5, 0.5, 0.001, 1)


@pytest.mark.gpu
@pytest.mark.parametrize("threads", [256, 128, 64])
def test_soft_nms_nan_inf_scores_and_lane_count_boundaries(ops, oracle, threads):
    """NaN / +-inf scores (a NaN score only ever wins as box i itself: cpu_nms.pyx:116-130 starts the running
    maximum from it and `maxscore < x` is false for NaN on either side), ties, duplicates, every method, ragged
    counts, sizes around the lane-count boundaries -- at every workgroup width the kernel is built for."""
    from simpledet_amd._lib import lib
    rs = np.random.RandomState(9)
    d = synth.nms_dets(3, 400)
    d[:, 4] = np.round(d[:, 4] * 8) / 8
    d[100:150] = d[:50]
    weird = synth.nms_dets(6, 400, mode="clustered")
    weird[[5, 77, 300], 4] = np.nan
    weird[[9, 200], 4] = np.inf
    weird[[11], 4] = -np.inf
    dets = np.stack([d, synth.nms_dets(4, 400, mode="all_overlap"), weird])
    lib().set_tuning("soft_nms_threads", threads)
    try:
        for method in (0, 1, 2):
            _soft_check(ops, oracle, dets, np.array([400, 400, 400]), 0.5, 0.3, 0.01, method)
        for Nmax in (1, 63, 64, 65, 127, 129, 255, 256, 257, 511, 513, 1000, 1025, 2049):
            b = np.stack([synth.nms_dets(70 + Nmax + p, Nmax) for p in range(2)])
            _soft_check(ops, oracle, b, np.array([Nmax, int(rs.randint(0, Nmax + 1))]), 0.5, 0.5, 0.001, 1)
    finally:
        lib().set_tuning("soft_nms_threads", 256)


@pytest.mark.gpu
def test_soft_nms_matches_reference_cython_fixture(ops):
    g = np.load(os.path.join(GOLD, "cython_nms.npz"))
    for seed in range(3):
        d = g["dets%d" % seed]
        for m, name in ((0, "hard"), (1, "linear"), (2, "gaussian")):
            od, oi, oc = ops.soft_nms_batched(_t(d[None]), None, 0.5, 0.3, 0.05, m)
            n = int(oc[0])
            np.testing.assert_array_equal(oi.cpu().numpy()[0, :n], g["soft_%s_inds%d" % (name, seed)])
            np.testing.assert_array_equal(od.cpu().numpy()[0, :n], g["soft_%s_boxes%d" % (name, seed)])
        ov = ops.bbox_overlaps(_t(d[:, :4]), _t(d[:40, :4]))
        np.testing.assert_array_equal(ov.cpu().numpy(), g["overlaps%d" % seed])


# ----------------------------------------------------------------------------- ProposalTarget ---
def _pt_check(ops, oracle, rois, gt, seed=1, **kw):
    B = rois.shape[0]
    p = oracle.make_pt_param(kw.pop("num_classes", 81), B, kw.pop("image_rois", 512), **kw)
    rng = oracle.GlibcRand(seed)
    want = oracle.proposal_target(rois, gt, p, rng=rng)
    state = ops.glibc_rand_state(seed)
    got = ops.proposal_target(
        _t(rois), _t(gt), p.num_classes, B, p.image_rois, p.fg_fraction, p.fg_thresh,
        p.bg_thresh_hi, p.bg_thresh_lo, bool(p.proposal_without_gt), bool(p.class_agnostic),
        tuple(p.bbox_mean), tuple(p.bbox_std), tuple(p.bbox_weight), rng_state=state,
        return_index=True)
    ro, lb, bt, bw, iou, kept = [x.cpu().numpy() for x in got]
    np.testing.assert_array_equal(kept, want[5])     # sampled indices: bit exact
    np.testing.assert_array_equal(ro, want[0])
    np.testing.assert_array_equal(lb, want[1])
    np.testing.assert_array_equal(iou, want[4])
    np.testing.assert_array_equal(bw, want[3])
    np.testing.assert_array_equal(bt != 0, want[2] != 0)
    np.testing.assert_allclose(bt, want[2], rtol=1e-6, atol=1e-7)
    # the generator state advanced by exactly the draws the reference consumes
    np.testing.assert_array_equal(state.cpu().numpy(), rng.state_words())
    return want


@pytest.mark.gpu
def test_proposal_target_baseline_config(ops, oracle):
    rois, gt = synth.proposal_target_inputs(0, 2, 2000, 100)
    want = _pt_check(ops, oracle, rois, gt)
    assert want[6] == 0
    assert (want[1] > 0).sum() > 0


@pytest.mark.gpu
def test_proposal_target_fuzz(ops, oracle):
    """120 random problems (8..700 proposals, 1..12 gt boxes, batch 1..3, sample sizes from 4 to 256,
    thresholds that empty or flood the fg / bg lists): every output and the glibc generator state
    against the oracle (itself pinned to the reference's compiled SampleROI)."""
    rs = np.random.RandomState(2024)
    for it in range(120):
        B = int(rs.randint(1, 4))
        num = int(rs.choice([8, 20, 64, 150, 333, 700]))
        ngt = tuple(int(x) for x in rs.randint(1, 13, B))
        rois, gt = synth.proposal_target_inputs(1000 + it, B, num, 16, n_gt=ngt)
        kw = dict(image_rois=int(rs.choice([4, 16, 64, 128, 256])),
                  fg_fraction=float(rs.choice([0.25, 0.5, 0.75])),
                  fg_thresh=float(rs.choice([0.1, 0.5, 0.7])),
                  bg_thresh_lo=float(rs.choice([0.0, 0.1])),
                  proposal_without_gt=bool(rs.randint(0, 2)),
                  class_agnostic=bool(rs.randint(0, 2)))
        kw["bg_thresh_hi"] = float(min(kw["fg_thresh"], rs.choice([0.3, 0.5])))
        if kw["class_agnostic"]:
            kw["num_classes"] = 2
        try:
            _pt_check(ops, oracle, rois, gt, seed=int(rs.randint(1, 1 << 30)), **kw)
        except AssertionError as e:
            raise AssertionError("problem %d %r: %s" % (it, kw, e))


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["few_fg", "many_fg", "no_neg_pad", "agnostic", "without_gt",
                                  "tiny", "state_carries"])
def test_proposal_target_cases(ops, oracle, case):
    if case == "few_fg":       # fg list shorter than fg_per_image: no fg shuffle, bg shuffle only
        rois, gt = synth.proposal_target_inputs(1, 2, 500, 20, n_gt=(2, 1))
        _pt_check(ops, oracle, rois, gt, image_rois=128)
    elif case == "many_fg":    # low fg threshold: fg shuffle + truncation
        rois, gt = synth.proposal_target_inputs(2, 2, 800, 50, n_gt=(30, 40))
        _pt_check(ops, oracle, rois, gt, image_rois=64, fg_thresh=0.2, bg_thresh_hi=0.2)
    elif case == "no_neg_pad":  # bg window excludes most rois: the neg padding loop runs (repeatedly)
        rois, gt = synth.proposal_target_inputs(3, 2, 60, 10, n_gt=(3, 4))
        _pt_check(ops, oracle, rois, gt, image_rois=256, bg_thresh_lo=0.1)
    elif case == "agnostic":
        rois, gt = synth.proposal_target_inputs(4, 3, 400, 30)
        _pt_check(ops, oracle, rois, gt, image_rois=128, class_agnostic=True, num_classes=2,
                  bbox_mean=(0.01, -0.02, 0.03, 0.0), bbox_weight=(1, 2, 3, 4))
    elif case == "without_gt":
        rois, gt = synth.proposal_target_inputs(5, 2, 400, 30)
        _pt_check(ops, oracle, rois, gt, image_rois=128, proposal_without_gt=True)
    elif case == "tiny":
        rois, gt = synth.proposal_target_inputs(6, 1, 8, 3, n_gt=(1,))
        _pt_check(ops, oracle, rois, gt, image_rois=16, seed=12345)
    else:                       # two consecutive calls share one generator state
        rois, gt = synth.proposal_target_inputs(7, 2, 600, 40)
        import torch
        p = oracle.make_pt_param(81, 2, 128)
        rng = oracle.GlibcRand(1)
        state = ops.glibc_rand_state(1)
        for _ in range(2):
            want = oracle.proposal_target(rois, gt, p, rng=rng)
            got = ops.proposal_target(_t(rois), _t(gt), 81, 2, 128, rng_state=state,
                                      return_index=True)
            np.testing.assert_array_equal(got[5].cpu().numpy(), want[5])
        np.testing.assert_array_equal(state.cpu().numpy(), rng.state_words())


# ------------------------------------------------------------------------ Proposal_v3 (8(f)) ----
def test_proposal_v3_anchor_known_answer(oracle):
    # float/floor anchors agree with GenAnchor for the reference's settings
    a = oracle.proposal_v3_anchors(4, [8], [0.5, 1, 2])
    np.testing.assert_array_equal(a, [[-22, -10, 25, 13], [-14, -14, 17, 17], [-10, -22, 13, 25]])
    for stride in (4, 8, 16, 32, 64):
        np.testing.assert_array_equal(oracle.proposal_v3_anchors(stride, [8], [0.5, 1, 2]),
                                      oracle.gen_base_anchors(stride, [8], [0.5, 1, 2]).astype(np.float32))


@pytest.mark.gpu
@pytest.mark.parametrize("cfg", [
    dict(A=3, H=50, W=84, stride=16, pre=2000, post=2000, train=False),
    dict(A=3, H=50, W=84, stride=16, pre=2000, post=1000, train=True),
    dict(A=3, H=100, W=167, stride=8, pre=2000, post=2000, train=False),
    dict(A=15, H=50, W=84, stride=16, pre=6000, post=300, train=False, scales=(2, 4, 8, 16, 32)),
    dict(A=3, H=13, W=21, stride=64, pre=2000, post=2000, train=False),   # count < pre
    dict(A=3, H=13, W=21, stride=64, pre=-1, post=100, train=True),
    dict(A=3, H=25, W=42, stride=32, pre=1000, post=1500, train=True, thr=0.3, min_size=64)])
@pytest.mark.parametrize("topk", [0, 1, 2])  # by level size / single workgroup / multi-workgroup
def test_proposal_v3_matches_oracle(ops, oracle, cfg, topk):
    from simpledet_amd._lib import lib
    scales = cfg.get("scales", (8,))
    ratios = (0.5, 1.0, 2.0)
    cls, bb, info = synth.rpn_outputs(3, 2, cfg["A"], cfg["H"], cfg["W"], cfg["stride"])
    thr, ms = cfg.get("thr", 0.7), cfg.get("min_size", 16)
    want = oracle.proposal_v3(cls, bb, info, cfg["pre"], cfg["post"], thr, ms, scales, ratios,
                              cfg["stride"], cfg["train"])
    lib().set_tuning("proposal_topk", topk)
    try:
        out, score = ops.proposal_v3(_t(cls), _t(bb), _t(info), cfg["pre"], cfg["post"], thr, ms,
                                     scales, ratios, cfg["stride"], cfg["train"])
    finally:
        lib().set_tuning("proposal_topk", 0)
    np.testing.assert_array_equal(score.cpu().numpy(), want[1])
    np.testing.assert_array_equal(out.cpu().numpy(), want[0])


@pytest.mark.gpu
@pytest.mark.parametrize("topk", [1, 2])
def test_proposal_v3_score_ties_are_stable(ops, oracle, topk):
    from simpledet_amd._lib import lib
    cls, bb, info = synth.rpn_outputs(5, 1, 3, 50, 84, 16)
    cls[:, 3:] = np.round(cls[:, 3:] * 32) / 32   # thousands of exactly tied scores
    want = oracle.proposal_v3(cls, bb, info, 1000, 1000, 0.7, 0, (8,), (0.5, 1, 2), 16, False)
    lib().set_tuning("proposal_topk", topk)
    try:
        out, score = ops.proposal_v3(_t(cls), _t(bb), _t(info), 1000, 1000, 0.7, 0, (8,),
                                     (0.5, 1, 2), 16)
    finally:
        lib().set_tuning("proposal_topk", 0)
    np.testing.assert_array_equal(score.cpu().numpy(), want[1])
    np.testing.assert_array_equal(out.cpu().numpy(), want[0])


def _top_proposal_cases():
    """tests/golden/top_proposal.npz: outputs of the reference's GetTopProposalOperator
    (models/FPN/get_top_proposal.py:10-43) run on a numpy shim of mx.nd (make_golden.py); inputs
    are regenerated from the seed with the generator's expressions."""
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", "top_proposal.npz"))
    for seed, B, N, top_n, q in g["cases"]:
        rs = np.random.RandomState(int(seed))
        bbox = (rs.rand(B, N, 4) * 800).astype(np.float32)
        score = rs.rand(B, N, 1).astype(np.float32)
        if q:
            score = (np.round(score * q) / q).astype(np.float32)
        yield int(seed), bbox, score, int(top_n), g["bbox_%d" % seed], g["score_%d" % seed]


def test_oracle_get_top_proposal_reproduces_reference_class(oracle):
    """includes tied scores (quantised): equal scores keep their input order, MXNet's stable
    descending SortByKey"""
    n = 0
    for seed, bbox, score, top_n, wb, ws_ in _top_proposal_cases():
        ob, os_ = oracle.get_top_proposal(bbox, score, top_n)
        np.testing.assert_array_equal(os_, ws_, err_msg="case %d" % seed)
        np.testing.assert_array_equal(ob, wb, err_msg="case %d" % seed)
        n += 1
    assert n == 6


@pytest.mark.gpu
def test_hip_get_top_proposal_reproduces_reference_class(ops):
    for seed, bbox, score, top_n, wb, ws_ in _top_proposal_cases():
        ob, os_ = ops.get_top_proposal(_t(bbox), _t(score), top_n)
        np.testing.assert_array_equal(os_.cpu().numpy(), ws_, err_msg="case %d" % seed)
        np.testing.assert_array_equal(ob.cpu().numpy(), wb, err_msg="case %d" % seed)


@pytest.mark.gpu
def test_get_top_proposal(ops, oracle):
    rs = np.random.RandomState(0)
    bbox = rs.rand(2, 10000, 4).astype(np.float32) * 500
    score = rs.rand(2, 10000, 1).astype(np.float32)
    score[0, :200] = 0.5
    for top_n in (2000, 10000, 12000):
        wb, ws_ = oracle.get_top_proposal(bbox, score, top_n)
        ob, os_ = ops.get_top_proposal(_t(bbox), _t(score), top_n)
        np.testing.assert_array_equal(os_.cpu().numpy(), ws_)
        np.testing.assert_array_equal(ob.cpu().numpy(), wb)


# ------------------------------------------------- DecodeBBox + test-time filter (8(f) rank 2) --
def _head_outputs(seed, B=2, R=1000, K=81):
    rs = np.random.RandomState(seed)
    rois = synth.random_rois(seed, B, R, degenerate=False)
    deltas = (rs.standard_normal((B, R, 4 * K)) * 0.5).astype(np.float32)
    logit = rs.standard_normal((B, R, K)).astype(np.float32) * 3
    e = np.exp(logit - logit.max(-1, keepdims=True))
    score = (e / e.sum(-1, keepdims=True)).astype(np.float32)
    info = np.array([[800, 1333, 1.0], [736, 1233, 1.5]][:B], np.float32)
    return rois, deltas, score, info


@pytest.mark.gpu
@pytest.mark.parametrize("agnostic", [True, False])
@pytest.mark.parametrize("kind", ["xywh", "xyxy"])
def test_decode_bbox_bit_exact(ops, oracle, agnostic, kind):
    rois, deltas, _, info = _head_outputs(0, K=2 if agnostic else 81)
    want = oracle.decode_bbox(rois, deltas, info, (0.0, 0.01, 0.0, -0.02), (0.1, 0.1, 0.2, 0.2),
                              agnostic, kind == "xyxy")
    got = ops.decode_bbox(_t(rois), _t(deltas), _t(info), (0.0, 0.01, 0.0, -0.02),
                          (0.1, 0.1, 0.2, 0.2), agnostic, kind)
    np.testing.assert_array_equal(got.cpu().numpy(), want)


@pytest.mark.gpu
@pytest.mark.parametrize("shared_box", [False, True])
def test_test_time_pipeline_decode_filter_softnms(ops, oracle, shared_box):
    """bbox head output -> DecodeBBox -> per-class filter -> batched soft-NMS, all on the device,
    equals the reference's CPU sequence (decodebbox.cc -> detection_test.py do_nms -> soft_nms)."""
    B, R, K = 2, 300, 12
    rois, deltas, score, info = _head_outputs(1, B, R, K)
    if shared_box:
        boxes = oracle.decode_bbox(rois, deltas[:, :, :8].copy(), info, class_agnostic=True)
        tb = ops.decode_bbox(_t(rois), _t(deltas[:, :, :8].copy()), _t(info), class_agnostic=True)
    else:
        boxes = oracle.decode_bbox(rois, deltas, info, class_agnostic=False)
        tb = ops.decode_bbox(_t(rois), _t(deltas), _t(info), class_agnostic=False)
    wd, wc = oracle.det_filter(boxes, score, 0.02)
    dets, counts = ops.det_filter(tb, _t(score), 0.02)
    np.testing.assert_array_equal(counts.cpu().numpy(), wc)
    dn = dets.cpu().numpy()
    for p in range(B * K):
        np.testing.assert_array_equal(dn[p, :wc[p]], wd[p, :wc[p]])
    od, oi, oc = ops.soft_nms_batched(dets, counts, 0.5, 0.5, 0.001, 1)
    od, oi, oc = od.cpu().numpy(), oi.cpu().numpy(), oc.cpu().numpy()
    for p in range(B * K):
        wb, wi = oracle.soft_nms(wd[p, :wc[p]], 0.5, 0.5, 0.001, 1)
        assert oc[p] == len(wi)
        np.testing.assert_array_equal(od[p, :oc[p]], wb)
        np.testing.assert_array_equal(oi[p, :oc[p]], wi)
