"""Secondary measurements printed inside bench.py's JSON line under "ops" (rank 0, N=1 only):
every other row of the hot-path table at its baseline shape, each with its own
bounding figure (8(d)): HBM GB/s for the streaming ops, problems/s + pair-IoU/s for the NMS family
(on-chip / latency bound), TFLOP/s against the fp32 MFMA peak for the DCN GEMM, and the CPU oracle
timed on the same inputs next to it.  Inputs are resident in HBM before each timed region; timing is
HIP events on the launch stream (torch's current stream is the stream handed to the C ABI).
"""
import time

import numpy as np

PEAK_HBM_GBS = 8000.0
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense, MI355X_MICROARCH.md
PEAK_F32_MFMA_TFLOPS = 157.3


def _time_gpu(fn, iters=20, warm=3):
    import torch
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def _time_cpu(fn, min_s=0.5, max_iter=20):
    fn()
    t0 = time.perf_counter()
    n = 0
    while True:
        fn()
        n += 1
        dt = time.perf_counter() - t0
        if dt >= min_s or n >= max_iter:
            return dt * 1e3 / n


def run(seed=0, cpu=True, only=None):
    import torch
    from simpledet_amd import ops, synth
    res = {}
    orc = None
    if cpu:
        from oracle import pyoracle as orc

    def T(a):
        return torch.from_numpy(np.ascontiguousarray(a)).cuda()

    def want(name):  # `only`: iterable of section names (A/B runs), None = everything
        return only is None or name in only

    def N_(t):
        return t.detach().cpu().numpy()

    def same(a, b):
        return bool(np.array_equal(N_(a) if hasattr(a, "cpu") else a, b))

    def close(a, b, tol):
        a = N_(a) if hasattr(a, "cpu") else a
        return bool(float(np.abs(a - b).max()) <= tol)

    # ---- GenAnchor: P2..P6, A = 3 (pure write: 16 B per anchor) ----
    if want("gen_anchor"):
        shapes = list(synth.FPN_SHAPES) + [(13, 21)]
        strides = [4, 8, 16, 32, 64]
        ms = _time_gpu(lambda: [ops.gen_anchor(h, w, s, [8], [0.5, 1, 2]) for (h, w), s in zip(shapes, strides)])
        nbytes = sum(16 * h * w * 3 for h, w in shapes)
        res["gen_anchor"] = {"ms": ms, "bytes": nbytes, "GBs": nbytes / ms / 1e6, "launches": 5}
        ms1 = _time_gpu(lambda: ops.gen_anchor_levels(shapes, strides, [8], [0.5, 1, 2]))
        res["gen_anchor"].update({"one_launch_ms": ms1, "one_launch_GBs": nbytes / ms1 / 1e6})
        if orc:
            res["gen_anchor"]["cpu_ms"] = _time_cpu(
                lambda: [orc.gen_anchor(h, w, s, [8], [0.5, 1, 2]) for (h, w), s in zip(shapes, strides)])
            res["gen_anchor"]["one_launch_matches_oracle"] = all(
                same(g, orc.gen_anchor(h, w, st_, [8], [0.5, 1, 2]))
                for g, ((h, w), st_) in zip(ops.gen_anchor_levels(shapes, strides, [8], [0.5, 1, 2]), zip(shapes, strides)))
            res["gen_anchor"]["matches_oracle"] = all(
                same(ops.gen_anchor(h, w, s, [8], [0.5, 1, 2]), orc.gen_anchor(h, w, s, [8], [0.5, 1, 2]))
                for (h, w), s in zip(shapes, strides))

    # ---- ProposalTarget: B=2, 2000 proposals, 100 gt slots, 512 rois, 81 classes ----
    if want("proposal_target"):
        rois, gt = synth.proposal_target_inputs(seed, 2, 2000, 100)
        tr, tg = T(rois), T(gt)
        state = ops.glibc_rand_state(1)
        ms = _time_gpu(lambda: ops.proposal_target(tr, tg, 81, 2, 512, rng_state=state))
        res["proposal_target"] = {"ms": ms, "images_per_s": 2 / ms * 1e3}
        if orc:
            p = orc.make_pt_param(81, 2, 512)
            rng = orc.GlibcRand(1)
            res["proposal_target"]["cpu_ms"] = _time_cpu(lambda: orc.proposal_target(rois, gt, p, rng=rng))
            g = ops.proposal_target(tr, tg, 81, 2, 512, rng_state=ops.glibc_rand_state(1), return_index=True)
            w = orc.proposal_target(rois, gt, p, rng=orc.GlibcRand(1))
            res["proposal_target"]["matches_oracle"] = (same(g[5], w[5]) and same(g[0], w[0]) and same(g[1], w[1])
                                                        and close(g[2], w[2], 2e-5))

    # ---- ProposalMaskTarget: the same sampling + 128 fg RoIs/img rasterised to 28x28 (mask head) ----
    if want("proposal_mask_target"):
        rois, gt = synth.proposal_target_inputs(seed, 2, 2000, 100)
        polys = synth.gt_polys(seed, gt, max_len=2500)
        tr, tg, tp = T(rois), T(gt), T(polys)
        state = ops.glibc_rand_state(1)
        ms = _time_gpu(lambda: ops.proposal_mask_target(tr, tg, tp, 81, 2, 512, mask_size=28, rng_state=state))
        res["proposal_mask_target"] = {"ms": ms, "images_per_s": 2 / ms * 1e3,
                                       "config": "B=2, 2000 proposals, 512 RoIs (128 fg masks 28x28) per image"}
        if orc:
            p = orc.make_pt_param(81, 2, 512)
            rng = orc.GlibcRand(1)
            res["proposal_mask_target"]["cpu_ms"] = _time_cpu(
                lambda: orc.proposal_mask_target(rois, gt, polys, p, 28, rng=rng))
            g = ops.proposal_mask_target(tr, tg, tp, 81, 2, 512, mask_size=28, rng_state=ops.glibc_rand_state(1))
            w = orc.proposal_mask_target(rois, gt, polys, p, 28, rng=orc.GlibcRand(1))
            res["proposal_mask_target"]["matches_oracle"] = same(g[0], w[0]) and same(g[5], w[5])
        # output_ratio = true (mask scoring R-CNN): + two image-resolution rasters per foreground RoI
        ms_r = _time_gpu(lambda: ops.proposal_mask_target(tr, tg, tp, 81, 2, 512, mask_size=28, rng_state=state,
                                                          output_ratio=True))
        res["proposal_mask_target"]["with_mask_ratio_ms"] = ms_r
        if orc:
            g = ops.proposal_mask_target(tr, tg, tp, 81, 2, 512, mask_size=28, rng_state=ops.glibc_rand_state(1),
                                         output_ratio=True)
            w = orc.proposal_mask_target(rois, gt, polys, p, 28, rng=orc.GlibcRand(1), output_ratio=True)
            res["proposal_mask_target"]["mask_ratio_matches_oracle"] = same(g[6], w[7]) and same(g[5], w[5])

    # ---- RPN anchor targets: P2-P6, 267k anchors/img, <= 40 gt, 256 sampled (loader op in the reference) ----
    if want("rpn_anchor_target"):
        gtb = synth.gt_boxes(seed, 2, 100)
        im = np.array([[800, 1333, 1.0], [800, 1333, 1.0]], np.float32)
        cfg = dict(stride=(4, 8, 16, 32, 64), short=(200, 100, 50, 25, 13), long=(334, 167, 84, 42, 21),
                   scales=(8,), aspects=(0.5, 1.0, 2.0))
        prm = ops.rpn_target_param(**cfg)
        st = ops.mt19937_state(seed=0)
        ti, tg = T(im), T(gtb)
        ms = _time_gpu(lambda: ops.rpn_anchor_target(ti, tg, prm, st, layout=1))
        res["rpn_anchor_target"] = {"ms": ms, "images_per_s": 2 / ms * 1e3,
                                    "config": "B=2, P2-P6 267,069 anchors/img, 256 sampled, numpy MT19937 replay"}
        if cpu:
            from oracle import rpn_target as orpn
            c2 = dict(cfg, allowed_border=0, pos_thr=0.7, neg_thr=0.3, min_pos_thr=0.0, image_anchor=256,
                      pos_fraction=0.5)
            rs = np.random.RandomState(0)
            res["rpn_anchor_target"]["cpu_ms"] = _time_cpu(
                lambda: [orpn.rpn_target(im[i], gtb[i], c2, rs) for i in range(2)], max_iter=5)
            try:
                g = ops.rpn_anchor_target(ti, tg, prm, ops.mt19937_state(seed=0), layout=1)
                rs2 = np.random.RandomState(0)
                w = [orpn.rpn_target(im[i], gtb[i], c2, rs2) for i in range(2)]
                res["rpn_anchor_target"]["matches_oracle"] = all(
                    same(g[0][i].reshape(-1), np.asarray(w[i][0]).reshape(-1).astype(N_(g[0]).dtype)) for i in range(2))
            except Exception as e:  # layouts differ between the device op and the loader class: see the tests
                res["rpn_anchor_target"]["matches_oracle"] = "not checked here (%s); tests/test_rpn_target.py" % type(e).__name__

    # ---- _contrib_NMS: B=2 x 2000 boxes, thr 0.7, post 1000 (train proposals) ----
    if want("nms"):
        dets = np.stack([synth.nms_dets(seed + i, 2000) for i in range(2)])
        td = T(dets)
        ms = _time_gpu(lambda: ops.nms(td, 2000, 1000, 0.7))
        res["nms"] = {"ms": ms, "images_per_s": 2 / ms * 1e3, "pair_iou_per_s": 2 * 2000 * 1999 / 2 / ms * 1e3,
                      "config": "B=2, N=2000, post 1000, thr 0.7"}
        if orc:
            res["nms"]["cpu_ms"] = _time_cpu(lambda: orc.nms(dets, 2000, 1000, 0.7))
            g, w = ops.nms(td, 2000, 1000, 0.7), orc.nms(dets, 2000, 1000, 0.7)
            res["nms"]["matches_oracle"] = same(g[0], w[0]) and same(g[1].reshape(w[1].shape), w[1])
        # more rows than one LDS sort holds: radix select of the pre_nms_top_n best, then the sort
        big = np.stack([synth.nms_dets(seed + 7 + i, 60000) for i in range(2)])
        tb = T(big)
        res["nms"]["n60000_pre6000_ms"] = _time_gpu(lambda: ops.nms(tb, 6000, 1000, 0.7))
        if orc:
            g, w = ops.nms(tb, 6000, 1000, 0.7), orc.nms(big, 6000, 1000, 0.7)
            res["nms"]["n60000_matches_oracle"] = same(g[0], w[0]) and same(g[1].reshape(w[1].shape), w[1])

    # ---- Proposal_v3 over the five FPN levels + get_top_proposal ----
    if want("proposal_v3_fpn"):
        lv = [synth.rpn_outputs(seed + i, 2, 3, h, w, st) for i, ((h, w), st) in
              enumerate(zip(list(synth.FPN_SHAPES) + [(13, 21)], [4, 8, 16, 32, 64]))]
        tl = [(T(c), T(b), T(i)) for c, b, i in lv]

        def rpn_proposals():
            outs = [ops.proposal_v3(c, b, i, 2000, 2000, 0.7, 0, (8,), (0.5, 1, 2), st)
                    for (c, b, i), st in zip(tl, [4, 8, 16, 32, 64])]
            return ops.get_top_proposal(torch.cat([o[0] for o in outs], 1),
                                        torch.cat([o[1] for o in outs], 1), 2000)
        ms = _time_gpu(rpn_proposals, iters=10, warm=2)
        res["proposal_v3_fpn"] = {"ms": ms, "images_per_s": 2 / ms * 1e3,
                                  "config": "B=2, P2-P6, 267k anchors/img, pre/post 2000 per level + top 2000"}
        if orc:
            c, b, i = lv[2]
            t = _time_cpu(lambda: orc.proposal_v3(c, b, i, 2000, 2000, 0.7, 0, (8,), (0.5, 1, 2), 16),
                          min_s=0.3, max_iter=5)
            res["proposal_v3_fpn"]["cpu_ms_p4_level_only"] = t
            g = ops.proposal_v3(tl[2][0], tl[2][1], tl[2][2], 2000, 2000, 0.7, 0, (8,), (0.5, 1, 2), 16)
            w = orc.proposal_v3(c, b, i, 2000, 2000, 0.7, 0, (8,), (0.5, 1, 2), 16)
            res["proposal_v3_fpn"]["matches_oracle"] = same(g[0], w[0]) and same(g[1].reshape(w[1].shape), w[1])
            res["proposal_v3_fpn"]["matches_oracle_scope"] = "P4 level (the other levels and the chain: tests/)"

    # ---- the train-time chain on the device: Proposal_v3 x5 -> get_top_proposal -> ProposalTarget ->
    # fused RoIAlign fwd + bwd, and the mask branch (models/FPN/builder.py:259-324, 567-610); eager
    # launches and one HIP graph of the same chain (capturing it proves there is no host sync in it) ----
    if want("train_chain"):
        import sys, os
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests"))
        from test_train_chain import gpu_chain, _inputs, IMAGE_ROIS
        rpn, gt, polys, feats = _inputs(seed + 11, 256)
        t_rpn = [(T(c), T(b), T(i)) for c, b, i in rpn]
        t_gt, t_polys, t_feats = T(gt), T(polys), [T(f) for f in feats]
        t_dy = torch.randn((2, IMAGE_ROIS, 256, 7, 7), device="cuda")
        t_dy14 = torch.randn((2, IMAGE_ROIS // 4, 256, 14, 14), device="cuda")
        rng, rngm = ops.glibc_rand_state(1), ops.glibc_rand_state(1)
        run_chain = lambda: gpu_chain(ops, t_rpn, t_gt, t_polys, t_feats, t_dy, t_dy14, rng, rngm)
        ms_eager = _time_gpu(run_chain, iters=10, warm=3)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            run_chain()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            run_chain()
        ms_graph = _time_gpu(graph.replay, iters=10, warm=3)
        res["train_chain"] = {"train_chain_ms": ms_graph, "eager_ms": ms_eager, "images_per_s": 2 / ms_graph * 1e3,
                              "host_syncs_in_chain": 0,
                              "config": "B=2: Proposal_v3 P2-P6 (pre/post 2000) -> get_top_proposal 2000 -> "
                                        "ProposalTarget 512 -> fused RoIAlign 7x7 fwd+bwd (256 ch) + "
                                        "ProposalMaskTarget -> 14x14 RoIAlign fwd+bwd; one HIP graph"}
        del t_feats, t_dy, t_dy14, graph

    # ---- the TEST-time chain of one image (BASELINE configs[0]; detection_infer_speed.py:66-77 ->
    # detection_test.py:233-267): fused FPN RoIAlign forward of the 1000 test proposals -> [bbox head: the model's,
    # synthetic outputs here] -> DecodeBBox -> per-class score filter -> soft-NMS of every class, one HIP graph, no
    # host round trip; beside it what the reference runs after the head on the HOST: decodebbox.cc (CPU only),
    # the numpy filter and the Cython soft_nms, class after class (its Pool(cpu_count() // 2) spreads IMAGES) ----
    if want("test_chain"):
        B, R, K = 1, 1000, 81
        strides4 = [4, 8, 16, 32]
        rs = np.random.RandomState(seed + 31)
        feats = [torch.randn((B, 256, h, w), device="cuda") for h, w in synth.FPN_SHAPES]
        rois_np = synth.random_rois(seed + 31, B, R, degenerate=False)
        deltas_np = (rs.standard_normal((B, R, 4 * K)) * 0.5).astype(np.float32)
        logit = rs.standard_normal((B, R, K)).astype(np.float32) * 3
        e = np.exp(logit - logit.max(-1, keepdims=True))
        score_np = (e / e.sum(-1, keepdims=True)).astype(np.float32)
        info_np = np.array([[800, 1333, 1.0]], np.float32)
        t_rois, t_deltas, t_score, t_info = T(rois_np), T(deltas_np), T(score_np), T(info_np)
        MIN_SCORE = 0.05   # config/faster_r50v1_fpn_1x.py:177

        def chain():
            pooled, _ = ops.fpn_roi_align_forward_packed(feats, t_rois, strides4, (7, 7))
            boxes = ops.decode_bbox(t_rois, t_deltas, t_info, class_agnostic=False)
            dets, counts = ops.det_filter(boxes, t_score, MIN_SCORE)
            return (pooled,) + tuple(ops.soft_nms_batched(dets, counts, 0.5, 0.5, 0.001, 1)) + (counts,)
        ms_eager = _time_gpu(chain, iters=10, warm=3)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            chain()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            g_out = chain()
        ms_graph = _time_gpu(graph.replay, iters=10, warm=3)
        ms_roi = _time_gpu(lambda: ops.fpn_roi_align_forward_packed(feats, t_rois, strides4, (7, 7)), iters=10, warm=2)
        res["test_chain"] = {"ms": ms_graph, "eager_ms": ms_eager, "roi_align_fwd_ms": ms_roi,
                             "images_per_s": 1e3 / ms_graph, "host_syncs_in_chain": 0,
                             "config": "B=1, R=1000, 81 classes: fused FPN RoIAlign 7x7 fwd (256 ch, packed) -> DecodeBBox "
                                       "(per class) -> score > 0.05 -> soft-NMS (linear, Nt 0.5, thr 0.001) of all classes; "
                                       "one HIP graph"}
        if orc:
            graph.replay()
            torch.cuda.synchronize()
            pooled, od, oi, oc, counts = [N_(x) for x in g_out]
            t0 = time.perf_counter()
            wboxes = orc.decode_bbox(rois_np, deltas_np, info_np, class_agnostic=False)
            wd, wc = orc.det_filter(wboxes, score_np, MIN_SCORE)
            t_dec = time.perf_counter() - t0
            okk = np.array_equal(counts, wc)
            t0 = time.perf_counter()
            for q in range(B * K):
                wb, wi = orc.soft_nms(wd[q, :wc[q]], 0.5, 0.5, 0.001, 1)
                okk = okk and oc[q] == len(wi) and np.array_equal(od[q, :oc[q]], wb) and np.array_equal(oi[q, :oc[q]], wi)
            t_nms = time.perf_counter() - t0
            wf = orc.fpn_roi_align_fwd([N_(f) for f in feats], rois_np, strides4, (7, 7), nthreads=8)
            okk = okk and np.array_equal(pooled, wf[0])
            res["test_chain"]["matches_oracle"] = bool(okk)
            res["test_chain"]["matches_oracle_scope"] = "pooled features, every class's kept boxes / indices / order"
            res["test_chain"]["cpu_port_post_head_ms"] = (t_dec + t_nms) * 1e3
            res["test_chain"]["boxes_over_threshold"] = int(wc.sum())
            try:  # the reference's own Cython soft_nms (oracle/_ref), the way do_nms() walks the classes
                from oracle._ref import cpu_nms as _ref_nms
                t0 = time.perf_counter()
                for q in range(B * K):
                    if wc[q]:
                        _ref_nms.soft_nms(np.ascontiguousarray(wd[q, :wc[q]]), np.float32(0.5), np.float32(0.5),
                                          np.float32(0.001), np.uint8(1))
                t_ref = time.perf_counter() - t0
                res["test_chain"]["cpu_ms"] = (t_dec + t_ref) * 1e3
                res["test_chain"]["cpu_ms_what"] = ("post-head work of one image on one host core: oracle decode + filter "
                                                    "(%.1f ms) + the reference's Cython soft_nms over the classes (%.1f ms)"
                                                    % (t_dec * 1e3, t_ref * 1e3))
                res["test_chain"]["post_head_gpu_ms"] = ms_graph - ms_roi
            except Exception:
                pass
        del feats, graph, g_out

    # ---- batched soft-NMS: 16 images x 80 classes x 1000 boxes (BASELINE configs[2]) ----
    if want("soft_nms"):
        P, n = 16 * 80, 1000
        sd = np.stack([synth.nms_dets(seed + 100 + i, n) for i in range(P)])  # 1,280 independent problems
        tsd = T(sd)
        ms = _time_gpu(lambda: ops.soft_nms_batched(tsd, None, 0.5, 0.5, 0.001, 1), iters=5, warm=1)
        res["soft_nms"] = {"ms": ms, "problems": P, "boxes": n, "problems_per_s": P / ms * 1e3,
                           "config": "1280 problems x 1000 boxes, linear, Nt 0.5, thr 0.001"}
        if orc:
            t = _time_cpu(lambda: orc.soft_nms(sd[0], 0.5, 0.5, 0.001, 1), min_s=0.3)
            res["soft_nms"]["cpu_ms_per_problem"] = t
            res["soft_nms"]["cpu_problems_per_s_1core"] = 1e3 / t
            od, oi, oc = [N_(x) for x in ops.soft_nms_batched(tsd, None, 0.5, 0.5, 0.001, 1)]
            okk = True
            for q in range(P):   # every problem (the C oracle needs ~1 ms each)
                wb, wi = orc.soft_nms(sd[q], 0.5, 0.5, 0.001, 1)
                okk = okk and oc[q] == len(wi) and np.array_equal(od[q, :oc[q]], wb) and np.array_equal(oi[q, :oc[q]], wi)
            res["soft_nms"]["matches_oracle"] = bool(okk)
            res["soft_nms"]["matches_oracle_scope"] = "all %d problems" % P
            try:  # the reference's own Cython soft_nms (oracle/_ref, built from /root/reference)
                from oracle._ref import cpu_nms as _ref_nms
                t0 = time.perf_counter()
                for q in range(2):
                    _ref_nms.soft_nms(sd[q], np.float32(0.5), np.float32(0.5), np.float32(0.001), np.uint8(1))
                res["soft_nms"]["reference_cython_ms_per_problem"] = (time.perf_counter() - t0) * 1e3 / 2
            except Exception:
                pass

    # ---- mask-head RoIAlign: fused FPN extractor, 14x14, 128 fg RoIs / image (BASELINE configs[4]) ----
    if want("roi_align_14x14"):
        feats = [torch.randn((2, 256, h, w), device="cuda") for h, w in synth.FPN_SHAPES]
        r14 = T(synth.random_rois(seed, 2, 128))
        strides4 = [4, 8, 16, 32]
        o14, am14 = ops.fpn_roi_align_forward_packed(feats, r14, strides4, (14, 14))
        dy14 = torch.randn_like(o14)
        shapes = [f.shape for f in feats]
        grads = [torch.empty_like(f) for f in feats]
        ms_f = _time_gpu(lambda: ops.fpn_roi_align_forward_packed(feats, r14, strides4, (14, 14)))
        ms_b = _time_gpu(lambda: ops.fpn_roi_align_backward_packed(dy14, r14, am14, shapes, strides4,
                                                                    d_feats=grads))
        alg = sum(4 * f.numel() for f in feats) + 16 * r14.shape[0] * r14.shape[1] + 3 * 4 * o14.numel()
        res["roi_align_14x14"] = {"fwd_ms": ms_f, "bwd_ms": ms_b, "algorithmic_bytes": alg,
                                  "fwd_frac": alg / ms_f / 1e6 / PEAK_HBM_GBS,
                                  "bwd_frac": alg / ms_b / 1e6 / PEAK_HBM_GBS,
                                  "config": "P2-P5 256ch 800x1333, N=2, 128 RoIs/img, 14x14, packed arg-max"}
        # the same extractor with fp16 feature maps and fp16 output (BASELINE configs[4] "14x14 mask
        # RoIAlign fp16"): its own algorithmic bytes = fp16 maps + rois + fp16 output + packed arg-max
        # for the forward; the backward here = to_fp32, the fp32 kernel, to_fp16 (the casts counted)
        feats16 = [f.half() for f in feats]
        o16, am16 = ops.fpn_roi_align_forward_packed_f16(feats16, r14, strides4, (14, 14))
        dy16 = dy14.half()
        grads16 = [torch.empty_like(f) for f in feats16]
        ms_f16 = _time_gpu(lambda: ops.fpn_roi_align_forward_packed_f16(feats16, r14, strides4, (14, 14)))
        ms_b16 = _time_gpu(lambda: ops.fpn_roi_align_backward_packed_f16(dy16, r14, am16, shapes, strides4,
                                                                        d_feats=grads16))
        alg16 = sum(2 * f.numel() for f in feats) + 16 * r14.shape[0] * r14.shape[1] + 2 * o16.numel() + am16[0].numel()
        res["roi_align_14x14_fp16"] = {"fwd_ms": ms_f16, "bwd_ms": ms_b16, "algorithmic_bytes_fwd": alg16,
                                       "fwd_frac": alg16 / ms_f16 / 1e6 / PEAK_HBM_GBS,
                                       "fwd_speedup_vs_fp32_io": ms_f / ms_f16,
                                       "matches_fp32_path": bool(torch.equal(o16, ops.fpn_roi_align_forward_packed(
                                           [f.float() for f in feats16], r14, strides4, (14, 14))[0].half())),
                                       "config": "P2-P5 256ch fp16 maps, N=2, 128 RoIs/img, 14x14, fp16 output, "
                                                 "packed arg-max; backward = casts around the fp32 kernel"}
        del feats16, o16, am16, dy16, grads16
        if orc:
            w14 = orc.fpn_roi_align_fwd([N_(f) for f in feats], N_(r14), strides4, (14, 14), nthreads=16)
            res["roi_align_14x14"]["matches_oracle"] = same(o14, w14[0])
            wb = orc.fpn_roi_align_bwd(N_(dy14), N_(r14), w14[1], w14[2], shapes, strides4, nthreads=16)
            res["roi_align_14x14"]["bwd_max_abs_err"] = max(float(np.abs(N_(g) - w).max()) for g, w in zip(grads, wb))
        del feats, o14, am14, dy14, grads

    # ---- single-level ROIAlign_v2, the C4 family (config/faster_r50v1c4_c5_512roi_1x.py:90-94):
    # data (2,1024,50,84), stride 16, 512 RoIs / image, 7x7, the reference's three fp32 outputs ----
    if want("roi_align_c4"):
        data = torch.randn((2, 1024, 50, 84), device="cuda")
        rc4 = T(synth.random_rois(seed, 2, 512))
        o, mx, my = ops.roi_align_v2_forward(data, rc4, (7, 7), 1 / 16.0)
        dyc = torch.randn_like(o)
        dxc = torch.empty_like(data)
        ms_f = _time_gpu(lambda: ops.roi_align_v2_forward(data, rc4, (7, 7), 1 / 16.0))
        ms_b = _time_gpu(lambda: ops.roi_align_v2_backward(dyc, rc4, mx, my, data.shape, 1 / 16.0, d_data=dxc))
        alg = 4 * data.numel() + 16 * rc4.shape[0] * rc4.shape[1] + 3 * 4 * o.numel()
        res["roi_align_c4"] = {"fwd_ms": ms_f, "bwd_ms": ms_b, "algorithmic_bytes": alg,
                               "fwd_frac": alg / ms_f / 1e6 / PEAK_HBM_GBS,
                               "bwd_frac": alg / ms_b / 1e6 / PEAK_HBM_GBS,
                               "config": "C4 (2,1024,50,84) stride 16, 512 RoIs/img, 7x7, float arg-max planes"}
        if orc:
            w = orc.roi_align_v2_fwd(N_(data), N_(rc4), (7, 7), 1 / 16.0, nthreads=16)
            res["roi_align_c4"]["matches_oracle"] = same(o, w[0]) and same(mx, w[1]) and same(my, w[2])
            wdx = orc.roi_align_v2_bwd(N_(dyc), w[1], w[2], tuple(data.shape))
            res["roi_align_c4"]["bwd_max_abs_err"] = float(np.abs(N_(dxc) - wdx).max())
        del data, o, mx, my, dyc, dxc

    # ---- ROIPooling_v1: C4 map (2,1024,50,84), 1024 rois, 7x7 ----
    if want("roi_pool_v1"):
        rs = np.random.RandomState(seed)
        data = torch.randn((2, 1024, 50, 84), device="cuda")
        r = synth.random_rois(seed, 1, 1024)[0]
        prois = T(np.concatenate([rs.randint(0, 2, (1024, 1)).astype(np.float32), r], 1))
        o, idx = ops.roi_pool_v1_forward(data, prois, (7, 7), 1 / 16.0)
        dy = torch.randn_like(o)
        dx = torch.empty_like(data)
        ms_f = _time_gpu(lambda: ops.roi_pool_v1_forward(data, prois, (7, 7), 1 / 16.0))
        ms_b = _time_gpu(lambda: ops.roi_pool_v1_backward(dy, prois, idx, data.shape, 1 / 16.0, d_data=dx))
        alg = 4 * data.numel() + 20 * 1024 + 2 * 4 * o.numel()
        res["roi_pool_v1"] = {"fwd_ms": ms_f, "bwd_ms": ms_b, "algorithmic_bytes": alg,
                              "fwd_GBs": alg / ms_f / 1e6, "fwd_frac": alg / ms_f / 1e6 / PEAK_HBM_GBS,
                              "bwd_frac": alg / ms_b / 1e6 / PEAK_HBM_GBS}
        if orc:
            w = orc.roi_pool_v1_fwd(N_(data), N_(prois), (7, 7), 1 / 16.0)
            res["roi_pool_v1"]["matches_oracle"] = same(o, w[0]) and same(idx, w[1])
        del data, o, idx, dy, dx

    # ---- DeformableConvolution: x (16,256,50,84), 3x3, dg 4, F 256 ----
    if want("deform_conv"):
        N, C, H, W, F = 16, 256, 50, 84, 256
        x = torch.randn((N, C, H, W), device="cuda")
        off = torch.randn((N, 72, H, W), device="cuda") * 2
        wt = torch.randn((F, C, 3, 3), device="cuda") * 0.05
        ms_i = _time_gpu(lambda: ops.deform_im2col(x, off, (3, 3), 1, 1, 1, 4), iters=20, warm=2)
        # the layer's forward = the col-free fused kernel (sd_deform_conv_fwd_nocol); the im2col + GEMM
        # forward (sd_deform_conv_fwd, what a training step that keeps col for its backward runs) beside it
        ms_f = _time_gpu(lambda: ops.deform_conv_forward(x, off, wt, 1, 1, 1, 4), iters=10, warm=2)
        ms_fu = _time_gpu(lambda: ops.deform_conv_forward(x, off, wt, 1, 1, 1, 4, keep_col=True), iters=10, warm=2)
        # the same layer with offsets of half a pixel (what a trained offset branch emits; sigma = 2 above is
        # the stress case: 15-row windows per tile, LDS bank conflicts of random taps)
        off_s = off * 0.25
        ms_fs = _time_gpu(lambda: ops.deform_conv_forward(x, off_s, wt, 1, 1, 1, 4), iters=10, warm=2)
        del off_s
        y = ops.deform_conv_forward(x, off, wt, 1, 1, 1, 4)
        dyc = torch.randn_like(y)
        grads = (torch.empty_like(x), torch.empty_like(off), torch.empty_like(wt))
        ms_b = _time_gpu(lambda: ops.deform_conv_backward(dyc, x, off, wt, 1, 1, 1, 4, grads=grads),
                         iters=6, warm=2)
        _, fws = ops.deform_conv_forward(x, off, wt, 1, 1, 1, 4, keep_col=True)
        ms_bc = _time_gpu(lambda: ops.deform_conv_backward(dyc, x, off, wt, 1, 1, 1, 4, grads=grads, fwd_ws=fws),
                          iters=6, warm=2)
        del fws
        colm = ops.deform_im2col(x, off, (3, 3), 1, 1, 1, 4)
        ms_c2i = _time_gpu(lambda: ops.deform_col2im(colm, off, x.shape, (3, 3), 1, 1, 1, 4), iters=10, warm=2)
        ms_crd = _time_gpu(lambda: ops.deform_col2im_coord(colm, x, off, (3, 3), 1, 1, 1, 4), iters=20, warm=2)
        del colm
        P_ = H * W
        im2col_bytes = N * (4 * (C + 72) * P_ + 4 * 9 * C * P_)
        flops = 2.0 * N * F * 9 * C * P_
        gemm_ms = max(ms_fu - ms_i, 1e-6)
        from simpledet_amd._lib import lib as _lib
        import ctypes as _ct
        _lib().cdll.sd_deform_conv_fwd_nocol_workspace_bytes.restype = _ct.c_size_t
        ws_fused = int(_lib().cdll.sd_deform_conv_fwd_nocol_workspace_bytes(N, C, H, W, F, 3, 3, 1, 1, 1, 4))
        ws_unfused = int(_lib().cdll.sd_deform_conv_workspace_bytes(N, C, H, W, 3, 3, 1, 1, 1))
        io_bytes = 4 * (x.numel() + off.numel() + wt.numel() + y.numel())
        res["deform_conv"] = {
            "fwd_kernel": "sd::dcn_fwd_fused_kernel (sampling fused into the GEMM, no col matrix)",
            "fwd_unfused_ms": ms_fu, "fwd_workspace_bytes": ws_fused, "fwd_unfused_workspace_bytes": ws_unfused,
            "fwd_io_bytes": io_bytes, "fwd_frac_of_f16_mfma_peak": 3.0 * flops / ms_f / 1e9 / PEAK_BF16_MFMA_TFLOPS,
            "im2col_ms": ms_i, "im2col_GBs": im2col_bytes / ms_i / 1e6,
            "im2col_frac": im2col_bytes / ms_i / 1e6 / PEAK_HBM_GBS,
            "col2im_ms": ms_c2i, "col2im_coord_ms": ms_crd,
            "fwd_ms": ms_f, "fwd_ms_offsets_sigma_0p5": ms_fs, "bwd_ms": ms_b, "bwd_with_forward_col_ms": ms_bc, "gemm_ms": gemm_ms, "gemm_TFLOPs": flops / gemm_ms / 1e9,
            # fp32 products as three fp16 MFMA terms (scaled hi/lo split): 3 x the flops on the f16 pipe
            "gemm_arith": "fp32 in/out, 3 f16 MFMA terms per product of a scaled hi/lo split (deform_gemm_split=2)",
            "gemm_frac_of_bf16_mfma_peak": 3.0 * flops / gemm_ms / 1e9 / PEAK_BF16_MFMA_TFLOPS,
            "gemm_vs_f32_mfma_peak": flops / gemm_ms / 1e9 / PEAK_F32_MFMA_TFLOPS,
            "config": "x (16,256,50,84), 3x3 pad 1, 4 deformable groups, 256 filters, fp32, offsets ~ N(0, 2^2)"}
        if orc:
            # every image of the batch against the oracle, north_star's ABSOLUTE bar: 1e-4 wherever the
            # values stay within |y| <= 32, scaled with the magnitude above that (tests/test_deform_conv.py)
            wy = orc.deform_conv_fwd(N_(x), N_(off), N_(wt), 1, 1, 1, 4)
            errs = np.abs(N_(y) - wy).reshape(N, -1).max(1)
            bar = 1e-4 * max(1.0, float(np.abs(wy).max()) / 32.0)
            res["deform_conv"]["fwd_max_abs_err"] = float(errs.max())
            res["deform_conv"]["fwd_abs_bar"] = bar
            res["deform_conv"]["matches_oracle"] = bool(float(errs.max()) <= bar)
            res["deform_conv"]["matches_oracle_scope"] = "forward of all %d images, absolute bar" % N
    return res
