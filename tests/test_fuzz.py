"""Randomised HIP-vs-oracle comparisons over shapes and parameters the fixed cases do not visit (GPU).
Each test draws its problems from a seeded generator, so a failure names a reproducible problem.
(The RNG-replaying ops have their own fuzz tests next to their cases: test_rpn_target.py,
test_detection_ops.py, test_mask_target.py, test_roi_align.py.)"""
import numpy as np
import pytest

from simpledet_amd import synth


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
def test_roi_align_v2_single_level_fuzz(ops, oracle):
    """ROIAlign_v2: 60 problems, pooled sizes 1x1 .. 9x9 and 14x14 (tiled kernels and the generic
    one), maps from 2x2 to 120x90, channel counts 1..40, scales 1/4 .. 1, RoIs beyond the image."""
    rs = np.random.RandomState(1)
    for it in range(60):
        B, C = int(rs.randint(1, 4)), int(rs.choice([1, 2, 7, 8, 16, 40]))
        H, W = int(rs.randint(2, 121)), int(rs.randint(2, 91))
        R = int(rs.choice([1, 5, 33, 64]))
        pooled = [(7, 7), (7, 7), (14, 14), (int(rs.randint(1, 10)), int(rs.randint(1, 10)))][int(rs.randint(0, 4))]
        scale = float(rs.choice([0.25, 0.125, 0.0625, 1.0]))
        data = rs.standard_normal((B, C, H, W)).astype(np.float32)
        rois = synth.random_rois(100 + it, B, R, int(H / scale) + 20, int(W / scale) + 20,
                                 degenerate=bool(rs.randint(0, 2)), min_size=2.0, max_size=float(max(H, W)) / scale)
        rois -= 10  # some boxes start left of / above the image
        msg = "problem %d data %s R=%d pooled=%s scale=%g" % (it, data.shape, R, pooled, scale)
        want = oracle.roi_align_v2_fwd(data, rois, pooled, scale, nthreads=8)
        got = ops.roi_align_v2_forward(_t(data), _t(rois), pooled, scale)
        for g, w, name in zip(got, want, ("output", "maxidx_x", "maxidx_y")):
            np.testing.assert_array_equal(g.cpu().numpy(), w, err_msg=msg + " " + name)
        dy = rs.standard_normal(want[0].shape).astype(np.float32)
        wdx = oracle.roi_align_v2_bwd(dy, want[1], want[2], data.shape, nthreads=8)
        gdx = ops.roi_align_v2_backward(_t(dy), _t(rois), got[1], got[2], data.shape, scale)[0].cpu().numpy()
        tol = 1e-4 * max(1.0, float(np.abs(wdx).max()) / 32.0)
        assert float(np.abs(gdx - wdx).max()) <= tol, msg + " backward"


@pytest.mark.gpu
def test_roi_pool_v1_fuzz(ops, oracle):
    """ROIPooling_v1: 60 problems; planes below and above the 64 KB LDS limit of the plane-resident
    kernel, more RoIs than one 256-RoI chunk, batch indices out of range, pooled 1x1 .. 14x14."""
    rs = np.random.RandomState(2)
    for it in range(60):
        B, C = int(rs.randint(1, 4)), int(rs.choice([1, 3, 8, 24]))
        H, W = [(5, 7), (50, 84), (13, 200), (150, 120), (1, 1), (64, 64)][int(rs.randint(0, 6))]
        K = int(rs.choice([1, 9, 255, 256, 257, 700]))
        pooled = [(7, 7), (7, 7), (14, 14), (int(rs.randint(1, 8)), int(rs.randint(1, 8)))][int(rs.randint(0, 4))]
        scale = float(rs.choice([0.0625, 0.125, 1.0]))
        data = rs.standard_normal((B, C, H, W)).astype(np.float32)
        box = synth.random_rois(200 + it, 1, K, int(H / scale) + 8, int(W / scale) + 8, degenerate=True,
                                min_size=1.0, max_size=float(max(H, W)) / scale)[0]
        bi = rs.randint(0, B, (K, 1)).astype(np.float32)
        if K > 4:
            bi[rs.randint(0, K, 2)] = [[-1.0], [float(B + 3)]]
        rois = np.concatenate([bi, box], 1).astype(np.float32)
        msg = "problem %d data %s K=%d pooled=%s scale=%g" % (it, data.shape, K, pooled, scale)
        ok = (bi[:, 0] >= 0) & (bi[:, 0] < B)
        safe = rois.copy()
        safe[~ok, 0] = 0  # the oracle (like the reference) would read out of bounds
        want, widx = oracle.roi_pool_v1_fwd(data, safe, pooled, scale)
        want[~ok], widx[~ok] = 0, -1
        out, idx = ops.roi_pool_v1_forward(_t(data), _t(rois), pooled, scale)
        np.testing.assert_array_equal(out.cpu().numpy(), want, err_msg=msg)
        np.testing.assert_array_equal(idx.cpu().numpy(), widx, err_msg=msg)
        dy = rs.standard_normal(want.shape).astype(np.float32)
        dy[~ok] = 0
        wdx = oracle.roi_pool_v1_bwd(dy, safe, widx, data.shape, scale)
        dx = ops.roi_pool_v1_backward(_t(dy), _t(rois), idx, data.shape, scale)[0].cpu().numpy()
        tol = 1e-5 * max(1.0, float(np.abs(wdx).max()))
        assert float(np.abs(dx - wdx).max()) <= tol, msg + " backward"


@pytest.mark.gpu
def test_gen_anchor_and_decode_bbox_fuzz(ops, oracle):
    rs = np.random.RandomState(3)
    for it in range(40):
        H, W = int(rs.randint(1, 120)), int(rs.randint(1, 200))
        stride = int(rs.choice([4, 8, 16, 32, 64]))
        scales = tuple(float(x) for x in rs.choice([2, 4, 8, 16, 32], int(rs.randint(1, 4)), replace=False))
        ratios = tuple(float(x) for x in rs.choice([0.25, 0.5, 1.0, 2.0, 3.0], int(rs.randint(1, 4)), replace=False))
        want = oracle.gen_anchor(H, W, stride, scales, ratios)
        got = ops.gen_anchor(H, W, stride, scales, ratios).cpu().numpy()
        np.testing.assert_array_equal(got.reshape(want.shape), want,
                                      err_msg="gen_anchor %d %dx%d s=%d %s %s" % (it, H, W, stride, scales, ratios))
    for it in range(40):
        B, R, K = int(rs.randint(1, 4)), int(rs.choice([1, 17, 300, 1000])), int(rs.choice([2, 21, 81]))
        rois = (rs.rand(B, R, 4) * 600).astype(np.float32)
        rois[..., 2:] += rois[..., :2]
        agn = bool(rs.randint(0, 2))
        pred = (rs.standard_normal((B, R, 4 * (2 if agn else K))) * 0.5).astype(np.float32)
        info = np.array([[rs.randint(300, 900), rs.randint(300, 1400), 1.0]] * B, np.float32)
        xyxy = bool(rs.randint(0, 2))
        want = oracle.decode_bbox(rois, pred, info, (0.0, 0.0, 0.0, 0.0), (0.1, 0.1, 0.2, 0.2), agn, xyxy)
        got = ops.decode_bbox(_t(rois), _t(pred), _t(info), (0.0, 0.0, 0.0, 0.0), (0.1, 0.1, 0.2, 0.2), agn,
                              "xyxy" if xyxy else "xywh")
        np.testing.assert_array_equal(got.cpu().numpy(), want, err_msg="decode_bbox %d agn=%d xyxy=%d" % (it, agn, xyxy))


@pytest.mark.gpu
def test_deform_conv_pieces_fuzz(ops, oracle):
    """deformable im2col / col2im / col2im_coord: 30 problems over kernel 1..3, pad 0..2, stride 1..2,
    dilation 1..2, 1..4 deformable groups, offsets up to several pixels (taps outside the map)."""
    rs = np.random.RandomState(4)
    done = 0
    for it in range(60):
        C = int(rs.choice([4, 8, 12]))
        dg = int(rs.choice([d for d in (1, 2, 4) if C % d == 0]))
        H, W = int(rs.randint(3, 30)), int(rs.randint(3, 40))
        k, pad, stride, dil = int(rs.randint(1, 4)), int(rs.randint(0, 3)), int(rs.randint(1, 3)), int(rs.randint(1, 3))
        Ho = (H + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
        Wo = (W + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
        if Ho < 1 or Wo < 1:
            continue
        done += 1
        x = rs.standard_normal((C, H, W)).astype(np.float32)
        off = (rs.standard_normal((dg * 2 * k * k, Ho, Wo)) * float(rs.choice([0.3, 1.5, 6.0]))).astype(np.float32)
        kw = dict(kernel=(k, k), pad=pad, stride=stride, dil=dil, dgroup=dg)
        a = dict(kernel=(k, k), pad=pad, stride=stride, dilate=dil, num_deformable_group=dg)
        msg = "problem %d x %s k=%d pad=%d stride=%d dil=%d dg=%d" % (it, x.shape, k, pad, stride, dil, dg)
        wc = oracle.deform_im2col(x, off, **kw)
        gc = ops.deform_im2col(_t(x[None]), _t(off[None]), **a).cpu().numpy()
        assert float(np.abs(gc.reshape(wc.shape) - wc).max()) <= 1e-4 * max(1.0, float(np.abs(wc).max())), msg + " im2col"
        dcol = rs.standard_normal(wc.shape).astype(np.float32)
        wx = oracle.deform_col2im(dcol, off, x.shape, **kw)
        gx = ops.deform_col2im(_t(dcol.reshape(gc.shape)), _t(off[None]), (1,) + x.shape, **a).cpu().numpy()
        assert float(np.abs(gx.reshape(wx.shape) - wx).max()) <= 1e-4 * max(1.0, float(np.abs(wx).max())), msg + " col2im"
        wo = oracle.deform_col2im_coord(dcol, x, off, **kw)
        go = ops.deform_col2im_coord(_t(dcol.reshape(gc.shape)), _t(x[None]), _t(off[None]), **a).cpu().numpy()
        assert float(np.abs(go.reshape(wo.shape) - wo).max()) <= 2e-4 * max(1.0, float(np.abs(wo).max())), msg + " col2im_coord"
    assert done >= 30


@pytest.mark.gpu
def test_proposal_v3_fuzz(ops, oracle):
    """Proposal_v3 (decode -> top-k -> NMS): 40 problems over map sizes 1x1 .. 120x170 (counts
    below one wave up to the multi-workgroup top-k), 1..15 anchors, pre / post above and below the
    candidate count, train / test filtering, min_size, tied scores; the three top-k strategies."""
    from simpledet_amd._lib import lib
    rs = np.random.RandomState(5)
    for it in range(40):
        A = int(rs.choice([1, 3, 9, 15]))
        scales = {1: (8,), 3: (8,), 9: (4, 8, 16), 15: (2, 4, 8, 16, 32)}[A]
        ratios = (1.0,) if A == 1 else (0.5, 1.0, 2.0)
        H, W = [(1, 1), (3, 5), (13, 21), (25, 42), (50, 84), (120, 170)][int(rs.randint(0, 6))]
        stride = int(rs.choice([8, 16, 32, 64]))
        B = int(rs.randint(1, 3))
        cls, bb, info = synth.rpn_outputs(300 + it, B, A, H, W, stride)
        if rs.randint(0, 3) == 0:
            cls[:, A:] = np.round(cls[:, A:] * 64) / 64
        pre = int(rs.choice([-1, 50, 1000, 6000]))
        post = int(rs.choice([1, 100, 300, 2000]))
        thr, ms, train = float(rs.choice([0.5, 0.7])), int(rs.choice([0, 16, 64])), bool(rs.randint(0, 2))
        topk = int(rs.randint(0, 3))
        msg = "problem %d A=%d %dx%d stride=%d B=%d pre=%d post=%d thr=%g min=%d train=%d topk=%d" % (
            it, A, H, W, stride, B, pre, post, thr, ms, train, topk)
        if pre <= 0 and A * H * W > 16384:
            # "all candidates" beyond the sort capacity is refused, not truncated (the reference's
            # configs use 2000 .. 12000)
            with pytest.raises(RuntimeError, match="exceeds 16384"):
                ops.proposal_v3(_t(cls), _t(bb), _t(info), pre, post, thr, ms, scales, ratios, stride, train)
            continue
        want = oracle.proposal_v3(cls, bb, info, pre, post, thr, ms, scales, ratios, stride, train)
        lib().set_tuning("proposal_topk", topk)
        try:
            out, score = ops.proposal_v3(_t(cls), _t(bb), _t(info), pre, post, thr, ms, scales, ratios,
                                         stride, train)
        finally:
            lib().set_tuning("proposal_topk", 0)
        np.testing.assert_array_equal(score.cpu().numpy(), want[1], err_msg=msg)
        np.testing.assert_array_equal(out.cpu().numpy(), want[0], err_msg=msg)
