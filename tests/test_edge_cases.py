"""Empty and ragged inputs through every entry point (GPU): nothing may crash, hang or write, and
the outputs keep the reference's shapes."""
import numpy as np
import pytest

from simpledet_amd import synth

STRIDES = [4, 8, 16, 32]


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


@pytest.mark.gpu
def test_empty_inputs(ops):
    import torch
    z = lambda *s: torch.zeros(s, device="cuda")  # noqa: E731
    # RoIAlign_v2: no rois / no channels
    out, mx, my = ops.roi_align_v2_forward(z(2, 4, 10, 12), z(2, 0, 4), (7, 7), 0.25)
    assert out.shape == (2, 0, 4, 7, 7) and mx.shape == out.shape
    dx = ops.roi_align_v2_backward(out, z(2, 0, 4), mx, my, (2, 4, 10, 12), 0.25)[0]
    assert dx.shape == (2, 4, 10, 12) and float(dx.abs().max()) == 0  # kWriteTo with nothing to add
    # fused FPN extractor, float and packed arg-max
    feats = [z(1, 8, h, w) for h, w in [(20, 24), (10, 12), (5, 6), (3, 3)]]
    o, mx, my = ops.fpn_roi_align_forward(feats, z(1, 0, 4), STRIDES, (7, 7))
    assert o.shape == (1, 0, 8, 7, 7)
    g = ops.fpn_roi_align_backward(o, z(1, 0, 4), mx, my, [f.shape for f in feats], STRIDES)
    assert all(float(t.abs().max()) == 0 for t in g)
    o, am = ops.fpn_roi_align_forward_packed(feats, z(1, 0, 4), STRIDES, (7, 7))
    g = ops.fpn_roi_align_backward_packed(o, z(1, 0, 4), am, [f.shape for f in feats], STRIDES)
    assert all(float(t.abs().max()) == 0 for t in g)
    # ROIPooling_v1
    o, ix = ops.roi_pool_v1_forward(z(2, 3, 8, 9), z(0, 5), (7, 7), 0.5)
    assert o.shape == (0, 3, 7, 7)
    dx, _ = ops.roi_pool_v1_backward(o, z(0, 5), ix, (2, 3, 8, 9), 0.5)
    assert float(dx.abs().max()) == 0
    # NMS / soft-NMS / IoU / top proposals
    out, score = ops.nms(z(2, 0, 5), 10, 5, 0.7)[:2]
    assert out.shape[0] == 2
    od, oi, oc = ops.soft_nms_batched(z(3, 0, 5), None, 0.5, 0.3, 0.001, 1)
    assert oc.shape == (3,) and int(oc.abs().max()) == 0
    od, oi, oc = ops.soft_nms_batched(z(0, 10, 5), None, 0.5, 0.3, 0.001, 1)
    assert oc.numel() == 0
    assert ops.bbox_overlaps(z(0, 4), z(5, 4)).shape == (0, 5)
    assert ops.bbox_overlaps(z(5, 4), z(0, 4)).shape == (5, 0)
    ob, os_ = ops.get_top_proposal(z(2, 0, 4), z(2, 0, 1), 7)
    assert ob.shape == (2, 7, 4) and float(ob.abs().max()) == 0 and float(os_.abs().max()) == 0
    # decode / filter
    b = ops.decode_bbox(z(2, 0, 4), z(2, 0, 8), torch.tensor([[100., 100., 1.]] * 2, device="cuda"),
                        (0, 0, 0, 0), (0.1, 0.1, 0.2, 0.2), class_agnostic=False)
    assert b.shape[1] == 0
    # DeformableConvolution with an empty batch
    y = ops.deform_conv_forward(z(0, 8, 9, 11), z(0, 72, 9, 11), z(6, 8, 3, 3), 1, 1, 1, 4)
    assert y.shape == (0, 6, 9, 11)
    # anchors of an empty map
    assert ops.gen_anchor(0, 5, 16, [8], [0.5, 1, 2]).numel() == 0


@pytest.mark.gpu
def test_proposal_target_without_ground_truth_or_foreground(ops, oracle):
    """All gt rows padded with -1 (no object in the image): every sampled roi is background."""
    rois, gt = synth.proposal_target_inputs(3, 2, 300, 20)
    gt[1, :, :] = -1
    want = oracle.proposal_target(rois, gt, oracle.make_pt_param(81, 2, 64), rng=oracle.GlibcRand(1))
    state = ops.glibc_rand_state(1)
    got = ops.proposal_target(_t(rois), _t(gt), 81, 2, 64, rng_state=state)
    for g, w in zip(got[:2], want[:2]):  # rois, labels
        np.testing.assert_array_equal(g.cpu().numpy().reshape(w.shape), w)


@pytest.mark.gpu
def test_fpn_roi_align_many_rois_per_image(ops, oracle):
    """3000 RoIs per image: long per-band RoI lists in the backward's LDS, many forward workgroups."""
    feats = synth.feature_maps(9, batch=1, channels=4)
    rois = synth.random_rois(9, 1, 3000)
    want = oracle.fpn_roi_align_fwd(feats, rois, STRIDES, (7, 7))
    tf = [_t(f) for f in feats]
    out, am = ops.fpn_roi_align_forward_packed(tf, _t(rois), STRIDES, (7, 7))
    np.testing.assert_array_equal(out.cpu().numpy(), want[0])
    o2, mx, my = ops.fpn_roi_align_forward(tf, _t(rois), STRIDES, (7, 7))
    np.testing.assert_array_equal(mx.cpu().numpy(), want[1])
    np.testing.assert_array_equal(my.cpu().numpy(), want[2])
    dy = np.random.RandomState(2).standard_normal(want[0].shape).astype(np.float32)
    wd = oracle.fpn_roi_align_bwd(dy, rois, want[1], want[2], [f.shape for f in feats], STRIDES)
    g1 = ops.fpn_roi_align_backward_packed(_t(dy), _t(rois), am, [f.shape for f in feats], STRIDES)
    g2 = ops.fpn_roi_align_backward(_t(dy), _t(rois), mx, my, [f.shape for f in feats], STRIDES)
    for a, b, w in zip(g1, g2, wd):
        s = max(1.0, float(np.abs(w).max()))
        assert np.abs(a.cpu().numpy() - w).max() <= 1e-4 * s
        assert np.abs(b.cpu().numpy() - w).max() <= 1e-4 * s


@pytest.mark.gpu
def test_nms_and_soft_nms_at_capacity(ops, oracle):
    """The largest sizes the in-LDS kernels accept: 16384 boxes for NMS, 4096 for soft-NMS."""
    d = synth.nms_dets(4, 16384)[None]
    want = oracle.nms(d, 6000, 1000, 0.7)
    got = ops.nms(_t(d), 6000, 1000, 0.7)
    np.testing.assert_array_equal(got[0].cpu().numpy(), want[0])
    np.testing.assert_array_equal(got[1].cpu().numpy(), want[1])
    s = synth.nms_dets(5, 4096)
    wk = oracle.soft_nms(s, 0.5, 0.3, 0.01, 1)
    od, oi, oc = ops.soft_nms_batched(_t(s[None]), None, 0.5, 0.3, 0.01, 1)
    n = int(oc[0])
    assert n == len(wk[0])
    np.testing.assert_array_equal(od[0, :n].cpu().numpy(), wk[0])
    np.testing.assert_array_equal(oi[0, :n].cpu().numpy(), wk[1])


@pytest.mark.gpu
@pytest.mark.parametrize("N,pre,post", [(20000, 6000, 1000), (70000, 12000, 2000), (16385, 16384, 300)])
def test_nms_more_rows_than_the_lds_sort_holds(ops, oracle, N, pre, post):
    """_contrib_NMS sorts any N (nms.cu:303) and keeps the first pre_nms_top_n rows of that order
    (:311-313).  Above 16384 unsorted rows the pre best are picked by a radix select and only they
    are sorted: same boxes, scores and keep indices as the oracle's full sort, ties included (a
    quarter of the scores are duplicated so that the selection threshold falls inside a tie run)."""
    d = synth.nms_dets(21, N)
    rs = np.random.RandomState(N)
    dup = rs.choice(N, N // 4, replace=False)
    d[dup, 4] = d[rs.choice(N, N // 4), 4]
    d = d[None]
    want = oracle.nms(d, pre, post, 0.7)
    got = ops.nms(_t(d), pre, post, 0.7, return_index=True)
    np.testing.assert_array_equal(got[0].cpu().numpy(), want[0])
    np.testing.assert_array_equal(got[1].cpu().numpy(), want[1])
    # the kept rows are rows of the input: original indices agree with a stable descending sort
    order = np.argsort(-d[0, :, 4], kind="stable")[:pre]
    idx = got[2].cpu().numpy()[0]
    assert set(idx[idx >= 0].tolist()) <= set(order.tolist())
    # every row unsorted and pre_nms_top_n beyond the sort capacity: a clear error, not a wrong answer
    with pytest.raises(RuntimeError, match="pre_nms_top_n <= 16384"):
        ops.nms(_t(d), 16385, post, 0.7)


@pytest.mark.gpu
def test_fused_fpn_roi_align_14x14_mask_head(ops, oracle):
    """The mask-head extractor (14x14, packed arg-max, strided bin order in the backward)."""
    feats = synth.feature_maps(12, batch=2, channels=6)
    rois = synth.random_rois(12, 2, 48)
    want = oracle.fpn_roi_align_fwd(feats, rois, STRIDES, (14, 14))
    tf = [_t(f) for f in feats]
    out, am = ops.fpn_roi_align_forward_packed(tf, _t(rois), STRIDES, (14, 14))
    np.testing.assert_array_equal(out.cpu().numpy(), want[0])
    np.testing.assert_array_equal(ops.argmax_codes(am[0], (14, 14)).cpu().numpy() == 255, want[1] == -1)
    dy = np.random.RandomState(5).standard_normal(want[0].shape).astype(np.float32)
    wd = oracle.fpn_roi_align_bwd(dy, rois, want[1], want[2], [f.shape for f in feats], STRIDES)
    gd = ops.fpn_roi_align_backward_packed(_t(dy), _t(rois), am, [f.shape for f in feats], STRIDES)
    o2, mx, my = ops.fpn_roi_align_forward(tf, _t(rois), STRIDES, (14, 14))
    g2 = ops.fpn_roi_align_backward(_t(dy), _t(rois), mx, my, [f.shape for f in feats], STRIDES)
    for a, b, w in zip(gd, g2, wd):
        s = max(1.0, float(np.abs(w).max()))
        assert np.abs(a.cpu().numpy() - w).max() <= 1e-4 * s
        assert np.abs(b.cpu().numpy() - w).max() <= 1e-4 * s


@pytest.mark.gpu
def test_signed_zero_scores_tie_like_the_reference(ops, oracle):
    """-0.0 and +0.0 compare equal in the reference's sorts (thrust::stable_sort_by_key with
    greater<float>, contrib/nms.cu:304-312; MXNet SortByKey behind mx.nd.argsort,
    models/FPN/get_top_proposal.py:26): zero-padded rows with either sign keep their input order."""
    rs = np.random.RandomState(11)
    N = 600
    bbox = (rs.rand(2, N, 4) * 500).astype(np.float32)
    score = rs.rand(2, N, 1).astype(np.float32)
    zero = rs.rand(2, N, 1) < 0.4
    sign = np.where(rs.rand(2, N, 1) < 0.5, np.float32(-0.0), np.float32(0.0))
    score = np.where(zero, sign, score).astype(np.float32)
    assert np.signbit(score).any() and (score == 0).sum() > 300
    for top_n in (100, 450, N):
        wb, ws_ = oracle.get_top_proposal(bbox, score, top_n)
        ob, os_ = ops.get_top_proposal(_t(bbox), _t(score), top_n)
        np.testing.assert_array_equal(ob.cpu().numpy(), wb)
        np.testing.assert_array_equal(os_.cpu().numpy().view(np.uint32), ws_.view(np.uint32))
    # _contrib_NMS: far-apart boxes so that nothing is suppressed and the output order is the sort's
    ctr = np.stack([np.arange(N) % 30 * 40.0, np.arange(N) // 30 * 40.0], 1)
    dets = np.concatenate([ctr, ctr + 10, np.zeros((N, 1))], 1)[None].repeat(2, 0).astype(np.float32)
    dets[..., 4:] = score
    wo, ws2, _ = oracle.nms(dets, N, N, 0.7)
    go, gs = ops.nms(_t(dets), rpn_pre_nms_top_n=N, rpn_post_nms_top_n=N, threshold=0.7)[:2]
    np.testing.assert_array_equal(go.cpu().numpy(), wo)
    np.testing.assert_array_equal(gs.cpu().numpy().view(np.uint32), ws2.view(np.uint32))
