"""ProposalMaskTarget's polygon rasteriser (convertPoly2Mask over the COCO mask API).

The COCO mask API is a third-party library the reference does not vendor (github.com/RogerChern/
cocoapi, doc/INSTALL.md:90-93), so no output of the real library exists here: PARITY UNPINNED.
oracle/mask_api.c restates the published pycocotools algorithm; these tests pin it by properties
that do not depend on the restatement, and the GPU tests compare the HIP rasteriser with the oracle
bit for bit on many polygons (the op-level cases are in tests/refcases.py)."""
import numpy as np
import pytest

from simpledet_amd import synth


def _poly_row(segs, L=200, cls=3):
    row = [float(cls), float(len(segs))] + [float(len(s)) for s in segs]
    for s in segs:
        row += [float(v) for v in s]
    out = np.full(L, -1.0, np.float32)
    out[:len(row)] = row
    return out


def test_axis_aligned_rectangle_fills_its_pixels(oracle):
    # RoI (0,0,28,28): w = h = 28 -> one mask pixel per image pixel.  Rectangle [4.5, 20.5) x [6.5, 10.5)
    roi = np.array([0, 0, 28, 28], np.float32)
    rect = [4.5, 6.5, 20.5, 6.5, 20.5, 10.5, 4.5, 10.5]  # x,y pairs
    m = oracle.poly2mask(roi, _poly_row([rect]), 28)
    want = np.zeros((28, 28), np.float32)
    want[7:11, 5:21] = 1  # pixel centres inside: rows 7..10, cols 5..20
    np.testing.assert_array_equal(m, want)
    # two disjoint rectangles are OR-ed; the same rectangle twice is still itself
    rect2 = [22.5, 20.5, 26.5, 20.5, 26.5, 25.5, 22.5, 25.5]
    m2 = oracle.poly2mask(roi, _poly_row([rect, rect2]), 28)
    want[21:26, 23:27] = 1
    np.testing.assert_array_equal(m2, want)
    np.testing.assert_array_equal(oracle.poly2mask(roi, _poly_row([rect, rect]), 28), m)


def test_mask_follows_the_roi_frame(oracle):
    # the same shape in a RoI twice as large occupies a quarter of the mask
    sq = [10, 10, 38, 10, 38, 38, 10, 38]
    a = oracle.poly2mask(np.array([10, 10, 38, 38], np.float32), _poly_row([sq]), 28)
    assert a.mean() > 0.9
    b = oracle.poly2mask(np.array([10, 10, 66, 66], np.float32), _poly_row([sq]), 28)
    assert abs(b.mean() - 0.25) < 0.05 and b[:13, :13].mean() > 0.9 and b[15:, 15:].sum() == 0
    # a polygon entirely outside the RoI leaves the mask empty
    c = oracle.poly2mask(np.array([100, 100, 128, 128], np.float32), _poly_row([sq]), 28)
    assert c.sum() == 0


@pytest.mark.parametrize("seed", range(6))
def test_area_and_mirror_symmetry(oracle, seed):
    rs = np.random.RandomState(seed)
    k = rs.randint(5, 20)
    ang = np.sort(rs.uniform(0, 2 * np.pi, k))
    px, py = 14 + 10 * np.cos(ang), 14 + 9 * np.sin(ang)  # convex, inside a (0,0,28,28) RoI
    roi = np.array([0, 0, 28, 28], np.float32)
    m = oracle.poly2mask(roi, _poly_row([np.stack([px, py], 1).reshape(-1)]), 28)
    area = 0.5 * abs(np.dot(px, np.roll(py, -1)) - np.dot(py, np.roll(px, -1)))
    per = np.hypot(np.diff(np.r_[px, px[0]]), np.diff(np.r_[py, py[0]])).sum()
    assert abs(m.sum() - area) <= 0.75 * per          # pixelisation error is bounded by the boundary
    assert set(np.unique(m)) <= {0.0, 1.0}
    # every row of a convex shape is one run
    for r in m:
        idx = np.flatnonzero(r)
        assert len(idx) == 0 or idx[-1] - idx[0] + 1 == len(idx)


# ------------------------------------------------------------------------------------------ GPU --
@pytest.mark.gpu
def test_hip_rasteriser_equals_oracle_on_many_polygons(ops, oracle):
    """One RoI per gt box (the box itself, jittered), fg_fraction 1: every row is rasterised."""
    import torch
    B, M = 2, 48
    gt = synth.gt_boxes(21, B, M, min_n=M, max_n=M)
    polys = synth.gt_polys(21, gt, max_len=400)
    rs = np.random.RandomState(3)
    rois = gt[:, :, :4] + rs.uniform(-3, 3, (B, M, 4)).astype(np.float32)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    got = ops.proposal_mask_target(t(rois), t(gt), t(polys), 81, B, 64, mask_size=28, fg_fraction=1.0,
                                   proposal_without_gt=True, rng_state=ops.glibc_rand_state(1),
                                   return_index=True)
    p = oracle.make_pt_param(81, B, 64, fg_fraction=1.0, proposal_without_gt=True)
    want = oracle.proposal_mask_target(rois, gt, polys, p, 28, rng=oracle.GlibcRand(1))
    np.testing.assert_array_equal(got[6].cpu().numpy(), want[6])
    np.testing.assert_array_equal(got[0].cpu().numpy(), want[0])
    np.testing.assert_array_equal(got[5].cpu().numpy(), want[5])
    m = got[5].cpu().numpy()
    assert (m == 1).any() and (m == -1).any()  # 48 < 64 rows: the tail keeps the -1 fill
    # mask sizes other than 28, long many-vertex polygons
    big = np.full((B, M, 3000), -1.0, np.float32)
    for b in range(B):
        for j in range(M):
            k = 700
            ang = np.linspace(0, 2 * np.pi, k, endpoint=False)
            x1, y1, x2, y2 = gt[b, j, :4]
            px = (x1 + x2) / 2 + (x2 - x1) * 0.45 * np.cos(ang) * (1 + 0.2 * np.sin(7 * ang))
            py = (y1 + y2) / 2 + (y2 - y1) * 0.45 * np.sin(ang) * (1 + 0.2 * np.cos(5 * ang))
            row = [gt[b, j, 4], 1.0, 2.0 * k] + list(np.stack([px, py], 1).reshape(-1))
            big[b, j, :len(row)] = row
    for ms in (14, 56):
        got = ops.proposal_mask_target(t(rois), t(gt), t(big), 81, B, 64, mask_size=ms, fg_fraction=1.0,
                                       proposal_without_gt=True, rng_state=ops.glibc_rand_state(1))
        want = oracle.proposal_mask_target(rois, gt, big, p, ms, rng=oracle.GlibcRand(1))
        np.testing.assert_array_equal(got[5].cpu().numpy(), want[5])


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def test_oracle_mask_ratio_properties(oracle):
    """convertPoly2MaskWithRatio (proposal_mask_target.cc:20-152): a polygon inside the RoI has all
    its pixels in the crop (ratio 1 up to the 1e-4 in the denominator), a RoI covering the left half
    of a rectangle about half, a RoI away from the polygon the 1e-10 floor; no segments -> the floor."""
    rect = lambda x1, y1, x2, y2: [x1, y1, x2, y1, x2, y2, x1, y2]
    poly = np.array([1.0, 1.0, 8.0] + rect(100, 50, 300, 150) + [-1.0] * 20, np.float32)
    m, r = oracle.poly2mask_ratio(np.array([90, 40, 310, 160], np.float32), poly, 28)
    assert abs(r - 1.0) < 1e-6 and m.sum() > 0
    m, r = oracle.poly2mask_ratio(np.array([100, 50, 200, 150], np.float32), poly, 28)
    assert 0.49 < r < 0.52
    m, r = oracle.poly2mask_ratio(np.array([400, 300, 500, 400], np.float32), poly, 28)
    assert r == 1e-10 and m.sum() == 0
    empty = np.array([1.0, 0.0] + [-1.0] * 20, np.float32)
    m, r = oracle.poly2mask_ratio(np.array([10, 10, 50, 50], np.float32), empty, 28)
    assert r == 1e-10 and m.sum() == 0


@pytest.mark.gpu
def test_hip_mask_ratio_equals_oracle(ops, oracle):
    """output_ratio = true: masks (double coordinates, :53-63) and ratios bit for bit, on RoIs that
    cut their polygon (jittered and shifted gt boxes), polygons of 1-3 possibly overlapping
    segments, and image-sized rasters; a raster above max_raster_pixels gives NaN for that row only."""
    B, M = 2, 40
    gt = synth.gt_boxes(31, B, M, min_n=M, max_n=M)
    polys = synth.gt_polys(31, gt, max_len=400)
    # overlapping segments: the second segment of every third polygon is a copy of the first, shifted
    for b in range(B):
        for j in range(0, M, 3):
            row = polys[b, j]
            if row[1] >= 2 and row[2] == row[3]:
                k = int(row[2])
                o = 2 + int(row[1])
                row[o + k:o + 2 * k] = row[o:o + k] + 2.5
    rs = np.random.RandomState(5)
    rois = gt[:, :, :4] + rs.uniform(-6, 6, (B, M, 4)).astype(np.float32)
    w = gt[:, :, 2] - gt[:, :, 0]
    rois[:, ::4, 0] += 0.3 * w[:, ::4]   # a quarter of the RoIs lose the left part of their object
    rois[:, ::4, 2] += 0.3 * w[:, ::4]
    rois = np.maximum(rois, 0).astype(np.float32)
    got = ops.proposal_mask_target(_t(rois), _t(gt), _t(polys), 81, B, 64, mask_size=28, fg_fraction=1.0,
                                   fg_thresh=0.3, proposal_without_gt=True, rng_state=ops.glibc_rand_state(1),
                                   return_index=True, output_ratio=True)
    p = oracle.make_pt_param(81, B, 64, fg_fraction=1.0, fg_thresh=0.3, proposal_without_gt=True)
    want = oracle.proposal_mask_target(rois, gt, polys, p, 28, rng=oracle.GlibcRand(1), output_ratio=True)
    np.testing.assert_array_equal(got[7].cpu().numpy(), want[6])      # kept index
    np.testing.assert_array_equal(got[5].cpu().numpy(), want[5])      # masks
    ratio = got[6].cpu().numpy()
    np.testing.assert_array_equal(ratio, want[7])
    live = want[7][want[7] != 0]
    assert len(live) > 40 and live.min() < 0.9 and live.max() > 0.99
    # the bound: rows whose crop or full raster has more pixels become NaN, the others stay
    small = ops.proposal_mask_target(_t(rois), _t(gt), _t(polys), 81, B, 64, mask_size=28, fg_fraction=1.0,
                                     fg_thresh=0.3, proposal_without_gt=True, rng_state=ops.glibc_rand_state(1),
                                     output_ratio=True, max_raster_pixels=20000)[6].cpu().numpy()
    nan = np.isnan(small)
    assert nan.any() and not nan.all()
    np.testing.assert_array_equal(small[~nan], want[7][~nan])


@pytest.mark.gpu
def test_proposal_target_v2_and_mask_target_fuzz(ops, oracle):
    """50 random problems each: ProposalTarget_v2 (valid_ranges, filter_scales on / off) and
    ProposalMaskTarget (sampling + masks) against the oracle: indices, labels, weights, IoU, masks
    and the glibc generator state bit for bit, box targets to 1e-6 (device logf)."""
    rs = np.random.RandomState(77)
    for it in range(50):
        B = int(rs.randint(1, 4))
        N = int(rs.choice([16, 100, 400]))
        ngt = tuple(int(x) for x in rs.randint(1, 9, B))
        rois, gt = synth.proposal_target_inputs(3000 + it, B, N, 12, n_gt=ngt)
        S = int(rs.choice([8, 32, 128]))
        fgf = float(rs.choice([0.25, 0.5]))
        lo = float(rs.choice([0, 32, 96]))
        vr = np.array([[lo, lo + float(rs.choice([64, 256, 2000]))]] * B, np.float32)
        filt = bool(rs.randint(0, 2))
        seed = int(rs.randint(1, 1 << 30))
        p = oracle.make_pt_param(81, B, S, fgf, 0.5, 0.5, 0.0, False)
        msg = "problem %d B=%d N=%d S=%d fg=%g ranges=%s filter=%d" % (it, B, N, S, fgf, vr[0].tolist(), filt)
        # --- ProposalTarget_v2
        rng = oracle.GlibcRand(seed)
        want = oracle.proposal_target(rois, gt, p, rng=rng, valid_ranges=vr, filter_scales=filt)
        st = ops.glibc_rand_state(seed)
        got = [x.cpu().numpy() for x in ops.proposal_target(
            _t(rois), _t(gt), 81, B, S, fgf, 0.5, 0.5, 0.0, False, False, rng_state=st,
            valid_ranges=_t(vr), filter_scales=filt, return_index=True)]
        np.testing.assert_array_equal(got[5], want[5], err_msg=msg)
        for k in (0, 1, 3, 4):
            np.testing.assert_array_equal(got[k], want[k], err_msg=msg)
        np.testing.assert_allclose(got[2], want[2], rtol=1e-6, atol=1e-7, err_msg=msg)
        np.testing.assert_array_equal(st.cpu().numpy(), rng.state_words(), err_msg=msg)
        # --- ProposalMaskTarget
        polys = synth.gt_polys(3000 + it, gt)
        rng = oracle.GlibcRand(seed)
        wm = oracle.proposal_mask_target(rois, gt, polys, p, 28, rng=rng, valid_ranges=vr, filter_scales=filt)
        st = ops.glibc_rand_state(seed)
        gm = [x.cpu().numpy() for x in ops.proposal_mask_target(
            _t(rois), _t(gt), _t(polys), 81, B, S, 28, fgf, 0.5, 0.5, 0.0, False, rng_state=st,
            valid_ranges=_t(vr), filter_scales=filt)]
        for k in (0, 1, 3, 4, 5):
            np.testing.assert_array_equal(gm[k], wm[k], err_msg=msg + " (mask target, output %d)" % k)
        np.testing.assert_allclose(gm[2], wm[2], rtol=1e-6, atol=1e-7, err_msg=msg)
        np.testing.assert_array_equal(st.cpu().numpy(), rng.state_words(), err_msg=msg)
