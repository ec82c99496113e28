"""Train-time proposal -> RoI-feature chain on the device (models/FPN/builder.py:259-324, 567-610,
symbol/builder.py:304): Proposal_v3 over P2-P6 -> concat -> get_top_proposal -> ProposalTarget ->
fused FPN RoIAlign forward / backward, and the mask branch (ProposalMaskTarget -> 14x14 RoIAlign).

Every operator is pinned on its own elsewhere; this checks that they COMPOSE -- each stage is fed
the previous stage's device output and compared with the oracle chain stage by stage (indices and
boxes bit for bit) -- and that the whole chain runs without a host synchronisation or a
device-to-host copy: it is captured into ONE HIP graph (stream capture rejects both) and replayed."""
import numpy as np
import pytest

from simpledet_amd import synth

STRIDES = list(synth.FPN_STRIDES)
RPN_STRIDES = [4, 8, 16, 32, 64]
RPN_SHAPES = list(synth.FPN_SHAPES) + [(13, 21)]
PRE = POST = 2000
NUM_CLASSES, IMAGE_ROIS = 81, 512


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _inputs(seed, channels):
    rpn = [synth.rpn_outputs(seed + i, 2, 3, h, w, st) for i, ((h, w), st) in
           enumerate(zip(RPN_SHAPES, RPN_STRIDES))]
    gt = synth.gt_boxes(seed + 7, 2, 100)
    polys = synth.gt_polys(seed + 8, gt, max_len=400)
    feats = synth.feature_maps(seed + 9, batch=2, channels=channels)
    return rpn, gt, polys, feats


def gpu_chain(ops, t_rpn, t_gt, t_polys, t_feats, t_dy, t_dy14, rng, rng_mask):
    """the device chain; every hand-off is a device tensor, nothing is read back"""
    import torch
    outs = [ops.proposal_v3(c, b, i, PRE, POST, 0.7, 0, (8,), (0.5, 1, 2), st, is_train=True)
            for (c, b, i), st in zip(t_rpn, RPN_STRIDES)]
    boxes = torch.cat([o[0] for o in outs], 1)
    scores = torch.cat([o[1] for o in outs], 1)
    top, top_s = ops.get_top_proposal(boxes, scores, POST)
    pt = ops.proposal_target(top, t_gt, NUM_CLASSES, 2, IMAGE_ROIS, rng_state=rng, return_index=True)
    rois = pt[0]
    out, am = ops.fpn_roi_align_forward_packed(t_feats, rois, STRIDES, (7, 7))
    grads = ops.fpn_roi_align_backward_packed(t_dy, rois, am, [f.shape for f in t_feats], STRIDES)
    # mask branch (models/maskrcnn/builder.py:115-134): its own sampling (the reference graph has one
    # or the other), then the 14x14 extractor on the sampled foreground RoIs
    mt = ops.proposal_mask_target(top, t_gt, t_polys, NUM_CLASSES, 2, IMAGE_ROIS, mask_size=28,
                                  rng_state=rng_mask, return_index=True)
    fg = mt[5].shape[1]
    mrois = mt[0][:, :fg].contiguous()
    out14, am14 = ops.fpn_roi_align_forward_packed(t_feats, mrois, STRIDES, (14, 14))
    grads14 = ops.fpn_roi_align_backward_packed(t_dy14, mrois, am14, [f.shape for f in t_feats], STRIDES)
    return {"levels": outs, "top": (top, top_s), "pt": pt, "out": out, "argmax": am, "grads": grads,
            "mt": mt, "out14": out14, "grads14": grads14}


@pytest.mark.gpu
def test_train_chain_composes_and_needs_no_host_sync(ops, oracle):
    import os
    import torch
    if os.environ.get("PYTORCH_NO_CUDA_MEMORY_CACHING") == "1":
        pytest.skip("stream capture forbids hipMalloc: the harness's allocations need the caching allocator")
    C = 32
    rpn, gt, polys, feats = _inputs(11, C)
    t_rpn = [(_t(c), _t(b), _t(i)) for c, b, i in rpn]
    t_gt, t_polys, t_feats = _t(gt), _t(polys), [_t(f) for f in feats]
    rs = np.random.RandomState(5)
    dy = rs.standard_normal((2, IMAGE_ROIS, C, 7, 7)).astype(np.float32)
    dy14 = rs.standard_normal((2, IMAGE_ROIS // 4, C, 14, 14)).astype(np.float32)
    t_dy, t_dy14 = _t(dy), _t(dy14)

    # ---- oracle chain, stage by stage on the oracle's own hand-offs ----
    o_levels = [oracle.proposal_v3(c, b, i, PRE, POST, 0.7, 0, (8,), (0.5, 1, 2), st, is_train=True)
                for (c, b, i), st in zip(rpn, RPN_STRIDES)]
    o_boxes = np.concatenate([o[0] for o in o_levels], 1)
    o_scores = np.concatenate([o[1] for o in o_levels], 1)
    o_top, o_top_s = oracle.get_top_proposal(o_boxes, o_scores, POST)
    par = oracle.make_pt_param(NUM_CLASSES, 2, IMAGE_ROIS)
    o_rng = oracle.GlibcRand(1)
    o_pt = oracle.proposal_target(o_top, gt, par, rng=o_rng)
    o_fwd = oracle.fpn_roi_align_fwd(feats, o_pt[0], STRIDES, (7, 7), nthreads=8)
    o_bwd = oracle.fpn_roi_align_bwd(dy, o_pt[0], o_fwd[1], o_fwd[2], [f.shape for f in feats], STRIDES,
                                     nthreads=8)
    o_rng_m = oracle.GlibcRand(1)
    o_mt = oracle.proposal_mask_target(o_top, gt, polys, par, 28, rng=o_rng_m)
    fg = o_mt[5].shape[1]
    o_mrois = np.ascontiguousarray(o_mt[0][:, :fg])
    o_fwd14 = oracle.fpn_roi_align_fwd(feats, o_mrois, STRIDES, (14, 14), nthreads=8)
    o_bwd14 = oracle.fpn_roi_align_bwd(dy14, o_mrois, o_fwd14[1], o_fwd14[2], [f.shape for f in feats],
                                       STRIDES, nthreads=8)

    def check(r, rng, rng_mask):
        for (gb, gs), (ob, osc), st in zip(r["levels"], o_levels, RPN_STRIDES):
            np.testing.assert_array_equal(gb.cpu().numpy(), ob, err_msg="Proposal_v3 boxes, stride %d" % st)
            np.testing.assert_array_equal(gs.cpu().numpy().reshape(osc.shape), osc,
                                          err_msg="Proposal_v3 scores, stride %d" % st)
        np.testing.assert_array_equal(r["top"][0].cpu().numpy(), o_top, err_msg="get_top_proposal boxes")
        np.testing.assert_array_equal(r["top"][1].cpu().numpy(), o_top_s, err_msg="get_top_proposal scores")
        pt = r["pt"]
        np.testing.assert_array_equal(pt[5].cpu().numpy(), o_pt[5], err_msg="ProposalTarget sampled rows")
        for k, name in ((0, "rois"), (1, "label"), (3, "bbox_weight"), (4, "match_gt_iou")):
            np.testing.assert_array_equal(pt[k].cpu().numpy(), o_pt[k], err_msg="ProposalTarget " + name)
        np.testing.assert_allclose(pt[2].cpu().numpy(), o_pt[2], rtol=2e-6, atol=2e-6)   # device logf
        np.testing.assert_array_equal(rng.cpu().numpy(), o_rng.state_words(),
                                      err_msg="rand() state after ProposalTarget")
        np.testing.assert_array_equal(rng_mask.cpu().numpy(), o_rng_m.state_words(),
                                      err_msg="rand() state after ProposalMaskTarget")
        np.testing.assert_array_equal(r["out"].cpu().numpy(), o_fwd[0], err_msg="RoIAlign forward")
        for g, w in zip(r["grads"], o_bwd):
            assert float(np.abs(g.cpu().numpy() - w).max()) <= 1e-4
        mt = r["mt"]
        np.testing.assert_array_equal(mt[6].cpu().numpy(), o_mt[6], err_msg="ProposalMaskTarget sampled rows")
        np.testing.assert_array_equal(mt[0].cpu().numpy(), o_mt[0], err_msg="ProposalMaskTarget rois")
        np.testing.assert_array_equal(mt[5].cpu().numpy(), o_mt[5], err_msg="mask targets")
        np.testing.assert_array_equal(r["out14"].cpu().numpy(), o_fwd14[0], err_msg="14x14 RoIAlign forward")
        for g, w in zip(r["grads14"], o_bwd14):
            assert float(np.abs(g.cpu().numpy() - w).max()) <= 1e-4

    # ---- eager run (also warms lazy initialisation up), compared with the oracle ----
    rng, rng_mask = ops.glibc_rand_state(1), ops.glibc_rand_state(1)
    eager = gpu_chain(ops, t_rpn, t_gt, t_polys, t_feats, t_dy, t_dy14, rng, rng_mask)
    torch.cuda.synchronize()
    check(eager, rng, rng_mask)

    # ---- the same chain captured into one HIP graph: stream capture fails on any host
    # synchronisation or synchronous device-to-host copy inside the window ----
    rng.copy_(ops.glibc_rand_state(1))
    rng_mask.copy_(ops.glibc_rand_state(1))
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        gpu_chain(ops, t_rpn, t_gt, t_polys, t_feats, t_dy, t_dy14, rng, rng_mask)  # allocator warm-up
    torch.cuda.current_stream().wait_stream(side)
    rng.copy_(ops.glibc_rand_state(1))
    rng_mask.copy_(ops.glibc_rand_state(1))
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        captured = gpu_chain(ops, t_rpn, t_gt, t_polys, t_feats, t_dy, t_dy14, rng, rng_mask)
    g.replay()
    torch.cuda.synchronize()
    check(captured, rng, rng_mask)
