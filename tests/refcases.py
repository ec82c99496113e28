"""Seeded cases on which the reference's own compiled operators (oracle/_ref/libref_*.so, built by
oracle/build_ref_cxx.py from /root/reference/operator_cxx where it lies), the CPU oracle and the
HIP kernels are compared.  One definition, three runners:

    run_case(name, "ref")     the reference's C++/CUDA-emulated operator      (container / _ref)
    run_case(name, "oracle")  oracle/liboracle.so                             (anywhere)
    run_case(name, "hip")     simpledet_amd through the C ABI                 (GPU box)

Each returns {output name: numpy array}.  tests/golden/make_golden_cxx.py stores the "ref" results
as SHA-256 digests (bit-exact ops; -0.0 canonicalised to +0.0) and, for the few float-tolerance
outputs, as arrays.  Inputs are regenerated from the seeds by simpledet_amd.synth, so only outputs
are stored.
"""
import hashlib

import numpy as np

from simpledet_amd import synth

STRIDES = list(synth.FPN_STRIDES)


def digest(a):
    a = np.ascontiguousarray(a)
    if a.dtype.kind == "f":
        a = a + a.dtype.type(0)  # -0.0 -> +0.0
    return hashlib.sha256(a.tobytes()).hexdigest()


def _t(a):
    import torch
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def _n(t):
    return t.detach().cpu().numpy()


def _ref(lib, name, **kw):
    from oracle import refmx
    return refmx.RefOp(lib, name, **kw)


def _orc():
    from oracle import pyoracle
    pyoracle.build()
    return pyoracle


def _ops():
    from simpledet_amd import ops
    return ops


# ------------------------------------------------------------------------------------ RoIAlign --
def _roi_align_level(runner, data, rois, stride, pooled=(7, 7)):
    if runner == "ref":
        op = _ref("roi_align_v2", "_contrib_ROIAlign_v2", pooled_size=pooled, spatial_scale=1.0 / stride)
        out, ax, ay = op.forward([data, rois], ctx="cpu")
    elif runner == "oracle":
        out, ax, ay = _orc().roi_align_v2_fwd(data, rois, pooled, 1.0 / stride, nthreads=8)
    else:
        out, ax, ay = [_n(t) for t in _ops().roi_align_v2_forward(_t(data), _t(rois), pooled, 1.0 / stride)]
    return {"out": out, "maxidx_x": ax, "maxidx_y": ay}


def case_roi_align_level(level, channels=16, pooled=(7, 7), num=512, seed=0):
    """ROIAlign_v2 on one pyramid level with ALL 1024 train-like RoIs (incl. the degenerate ones)."""
    def run(runner):
        feats = synth.feature_maps(seed, batch=2, channels=channels)
        rois = synth.random_rois(seed, 2, num)
        return _roi_align_level(runner, feats[level], rois, STRIDES[level], pooled)
    return run


def case_roi_align_c4(seed=1):
    """the single-level C4 family (symbol/builder.py:875-898) at reduced channels."""
    def run(runner):
        rs = np.random.RandomState(seed)
        data = rs.standard_normal((2, 32, 50, 84)).astype(np.float32)
        rois = synth.random_rois(seed, 2, 512)
        return _roi_align_level(runner, data, rois, 16)
    return run


def case_roi_align_fpn(channels, pooled=(7, 7), num=512, seed=0):
    """models/FPN/builder.py:588-605: fpn_roi_assign zeroes the RoIs of the other levels, one
    ROIAlign_v2 per level over all RoIs, add_n.  The fused op must equal that graph."""
    def run(runner):
        feats = synth.feature_maps(seed, batch=2, channels=channels)
        rois = synth.random_rois(seed, 2, num)
        if runner == "ref":
            per, _ = _orc().fpn_roi_assign(rois, STRIDES)  # pinned by tests/golden/fpn_assign.npz
            out = None
            for l, s in enumerate(STRIDES):
                op = _ref("roi_align_v2", "_contrib_ROIAlign_v2", pooled_size=pooled, spatial_scale=1.0 / s)
                o = op.forward([feats[l], per[l]], ctx="cpu")[0]
                out = o if out is None else out + o
        elif runner == "oracle":
            out = _orc().fpn_roi_align_fwd(feats, rois, STRIDES, pooled, nthreads=8)[0]
        else:
            out = _n(_ops().fpn_roi_align_forward([_t(f) for f in feats], _t(rois), STRIDES, pooled)[0])
        return {"out": out}
    return run


def case_roi_align_bwd(level, channels=4, num=96, seed=2, pooled=(7, 7)):
    """backward = the GPU scatter (roi_align_v2.cu:35-84), emulated serially in thread order."""
    def run(runner):
        feats = synth.feature_maps(seed, batch=2, channels=channels)
        data, stride = feats[level], STRIDES[level]
        rois = synth.random_rois(seed, 2, num)
        fwd = _roi_align_level("oracle", data, rois, stride, pooled)
        dy = np.random.RandomState(seed + 100).standard_normal(fwd["out"].shape).astype(np.float32)
        if runner == "ref":
            op = _ref("roi_align_v2", "_contrib_ROIAlign_v2", pooled_size=pooled, spatial_scale=1.0 / stride)
            dx, dr = op.backward([dy], [data, rois], [fwd["out"], fwd["maxidx_x"], fwd["maxidx_y"]], ctx="gpu")
        elif runner == "oracle":
            dx = _orc().roi_align_v2_bwd(dy, fwd["maxidx_x"], fwd["maxidx_y"], data.shape)
            dr = np.zeros_like(rois)
        else:
            dx, dr = _ops().roi_align_v2_backward(_t(dy), _t(rois), _t(fwd["maxidx_x"]), _t(fwd["maxidx_y"]),
                                                  data.shape, 1.0 / stride)
            dx, dr = _n(dx), _n(dr)
        return {"d_data": dx, "d_rois": dr}
    return run


# ------------------------------------------------------------------------------------- RoIPool --
def _pool_inputs(seed, C=16, K=128):
    rs = np.random.RandomState(seed)
    data = rs.standard_normal((2, C, 50, 84)).astype(np.float32)
    r = synth.random_rois(seed, 2, K // 2)
    rois = np.concatenate([np.repeat(np.arange(2), K // 2)[:, None].astype(np.float32), r.reshape(-1, 4)], 1)
    return data, rois


def case_roi_pool_fwd(ctx, seed=3):
    def run(runner):
        data, rois = _pool_inputs(seed)
        if runner == "ref":
            out, idx = _ref("roi_pooling_v1", "ROIPooling_v1", pooled_size=(7, 7),
                            spatial_scale=0.0625).forward([data, rois], ctx=ctx)
        elif runner == "oracle":
            out, idx = _orc().roi_pool_v1_fwd(data, rois, (7, 7), 0.0625)
        else:
            out, idx = [_n(t) for t in _ops().roi_pool_v1_forward(_t(data), _t(rois), (7, 7), 0.0625)]
        return {"out": out, "maxidx": idx}
    return run


def case_roi_pool_bwd(seed=3):
    def run(runner):
        data, rois = _pool_inputs(seed)
        out, idx = _orc().roi_pool_v1_fwd(data, rois, (7, 7), 0.0625)
        dy = np.random.RandomState(seed + 100).standard_normal(out.shape).astype(np.float32)
        if runner == "ref":
            dx = _ref("roi_pooling_v1", "ROIPooling_v1", pooled_size=(7, 7), spatial_scale=0.0625
                      ).backward([dy], [data, rois], [out, idx], ctx="gpu")[0]
        elif runner == "oracle":
            dx = _orc().roi_pool_v1_bwd(dy, rois, idx, data.shape, 0.0625)
        else:
            dx = _n(_ops().roi_pool_v1_backward(_t(dy), _t(rois), _t(idx), data.shape, 0.0625)[0])
        return {"d_data": dx}
    return run


# ----------------------------------------------------------------------------------- GenAnchor --
ANCHOR_CFGS = [(4, 200, 334, (8,), (0.5, 1, 2)), (8, 100, 167, (8,), (0.5, 1, 2)),
               (16, 50, 84, (8,), (0.5, 1, 2)), (32, 25, 42, (8,), (0.5, 1, 2)),
               (64, 13, 21, (8,), (0.5, 1, 2)), (16, 50, 84, (2, 4, 8, 16, 32), (0.5, 1, 2))]


def case_gen_anchor(i, ctx):
    stride, H, W, scales, ratios = ANCHOR_CFGS[i]

    def run(runner):
        if runner == "ref":
            op = _ref("generate_anchor", "_contrib_GenAnchor", scales=scales, ratios=ratios,
                      feature_stride=stride)
            a = op.forward([np.zeros((1, 2 * len(scales) * len(ratios), H, W), np.float32)], ctx=ctx)[0]
        elif runner == "oracle":
            a = _orc().gen_anchor(H, W, stride, list(scales), list(ratios))
        else:
            a = _n(_ops().gen_anchor(H, W, stride, list(scales), list(ratios)))
        return {"anchors": a}
    return run


# ------------------------------------------------------------------------------ ProposalTarget --
PT_CFGS = [dict(seed=0, B=2, N=2000, M=100, S=512), dict(seed=1, B=2, N=400, M=30, S=64),
           dict(seed=2, B=1, N=300, M=5, S=128), dict(seed=3, B=2, N=600, M=40, S=256, without_gt=True),
           dict(seed=4, B=2, N=500, M=20, S=128, fg_fraction=0.5, fg_thresh=0.6, bg_lo=0.1)]
PT_NAMES = ("roi_output", "label", "bbox_target", "bbox_weight", "match_gt_iou")


def case_proposal_target(i, class_agnostic=False, calls=1):
    c = PT_CFGS[i]

    def run(runner):
        kw = dict(num_classes=81, batch_images=c["B"], image_rois=c["S"],
                  fg_fraction=c.get("fg_fraction", 0.25), fg_thresh=c.get("fg_thresh", 0.5),
                  bg_thresh_hi=0.5, bg_thresh_lo=c.get("bg_lo", 0.0),
                  proposal_without_gt=c.get("without_gt", False), class_agnostic=class_agnostic)
        res = None
        if runner == "ref":
            from oracle import refmx
            op = _ref("proposal_target", "ProposalTarget", output_iou=True, **kw)
            refmx.srand(1)  # libc's state in a process that never called srand
            for k in range(calls):  # the state carries from call to call
                rois, gt = synth.proposal_target_inputs(c["seed"] + 10 * k, c["B"], c["N"], c["M"])
                res = op.forward([rois, gt])
        elif runner == "oracle":
            orc = _orc()
            rng = orc.GlibcRand(1)
            p = orc.make_pt_param(81, c["B"], c["S"], kw["fg_fraction"], kw["fg_thresh"], 0.5,
                                  kw["bg_thresh_lo"], kw["proposal_without_gt"], class_agnostic)
            for k in range(calls):
                rois, gt = synth.proposal_target_inputs(c["seed"] + 10 * k, c["B"], c["N"], c["M"])
                res = orc.proposal_target(rois, gt, p, rng=rng)[:5]
        else:
            ops = _ops()
            state = ops.glibc_rand_state(1)
            for k in range(calls):
                rois, gt = synth.proposal_target_inputs(c["seed"] + 10 * k, c["B"], c["N"], c["M"])
                res = [_n(t) for t in ops.proposal_target(_t(rois), _t(gt), rng_state=state, **kw)]
        return dict(zip(PT_NAMES, res))
    return run


PT2_CFGS = [dict(seed=5, B=2, N=1000, M=60, S=128, ranges=[[0, 90], [60, 400]], filter=True),
            dict(seed=6, B=2, N=600, M=40, S=128, ranges=[[0, 90], [60, 400]], filter=False),
            dict(seed=7, B=2, N=400, M=20, S=64, ranges=[[0, 1e5], [200, 300]], filter=True, without_gt=False,
                 fg_fraction=0.5)]


def case_proposal_target_v2(i):
    """ProposalTarget_v2 (tridentnet): gt boxes outside the image's valid scale range are not
    appended to the candidate rois."""
    c = PT2_CFGS[i]

    def run(runner):
        kw = dict(num_classes=81, batch_images=c["B"], image_rois=c["S"], fg_fraction=c.get("fg_fraction", 0.25),
                  fg_thresh=0.5, bg_thresh_hi=0.5, bg_thresh_lo=0.0, proposal_without_gt=c.get("without_gt", False),
                  class_agnostic=False)
        rois, gt = synth.proposal_target_inputs(c["seed"], c["B"], c["N"], c["M"])
        vr = np.array(c["ranges"], np.float32)
        if runner == "ref":
            from oracle import refmx
            op = _ref("proposal_target_v2", "ProposalTarget_v2", output_iou=True, filter_scales=c["filter"], **kw)
            refmx.srand(1)
            res = op.forward([rois, gt, vr])
        elif runner == "oracle":
            orc = _orc()
            p = orc.make_pt_param(81, c["B"], c["S"], kw["fg_fraction"], 0.5, 0.5, 0.0, kw["proposal_without_gt"])
            res = orc.proposal_target(rois, gt, p, rng=orc.GlibcRand(1), valid_ranges=vr,
                                      filter_scales=c["filter"])[:5]
        else:
            ops = _ops()
            res = [_n(t) for t in ops.proposal_target(_t(rois), _t(gt), rng_state=ops.glibc_rand_state(1),
                                                      valid_ranges=_t(vr), filter_scales=c["filter"], **kw)]
        return dict(zip(PT_NAMES, res))
    return run


PMT_CFGS = [dict(seed=8, B=2, N=1000, M=40, S=128), dict(seed=9, B=2, N=600, M=20, S=64, fg_fraction=0.5),
            dict(seed=10, B=2, N=500, M=30, S=128, ranges=[[0, 120], [80, 500]]),
            # output_ratio = true: the mask scoring R-CNN heads (models/msrcnn/builder.py:219-237)
            dict(seed=11, B=2, N=800, M=30, S=128, ratio=True),
            dict(seed=12, B=2, N=400, M=12, S=64, fg_fraction=0.5, ratio=True)]
PMT_NAMES = PT_NAMES + ("mask_target",)


def case_proposal_mask_target(i):
    """ProposalMaskTarget (Mask R-CNN / TridentNet): sampling + 28x28 masks of the gt polygons.  The
    "ref" runner is the reference's proposal_mask_target.cc compiled against the RESTATED COCO mask
    API (oracle/mask_api.c; the real one is not vendored): op logic pinned, rasterisation not."""
    c = PMT_CFGS[i]
    ratio = bool(c.get("ratio"))

    def run(runner):
        kw = dict(num_classes=81, batch_images=c["B"], image_rois=c["S"], fg_fraction=c.get("fg_fraction", 0.25),
                  fg_thresh=0.5, bg_thresh_hi=0.5, bg_thresh_lo=0.0, proposal_without_gt=False)
        rois, gt = synth.proposal_target_inputs(c["seed"], c["B"], c["N"], c["M"])
        polys = synth.gt_polys(c["seed"], gt)
        vr = np.array(c["ranges"], np.float32) if "ranges" in c else None
        if runner == "ref":
            from oracle import refmx
            op = _ref("proposal_mask_target", "ProposalMaskTarget", num_args=4 if vr is not None else 3,
                      mask_size=28, output_iou=True, filter_scales=vr is not None, output_ratio=ratio, **kw)
            refmx.srand(1)
            res = op.forward([rois, gt, polys] + ([vr] if vr is not None else []))
        elif runner == "oracle":
            orc = _orc()
            p = orc.make_pt_param(81, c["B"], c["S"], kw["fg_fraction"], 0.5, 0.5, 0.0, False)
            res = orc.proposal_mask_target(rois, gt, polys, p, 28, rng=orc.GlibcRand(1), valid_ranges=vr,
                                           filter_scales=vr is not None, output_ratio=ratio)
            res = list(res[:6]) + ([res[7]] if ratio else [])
        else:
            ops = _ops()
            res = [_n(t) for t in ops.proposal_mask_target(
                _t(rois), _t(gt), _t(polys), mask_size=28, rng_state=ops.glibc_rand_state(1),
                valid_ranges=None if vr is None else _t(vr), filter_scales=vr is not None,
                output_ratio=ratio, **kw)]
        return dict(zip(PMT_NAMES + (("mask_ratio",) if ratio else ()), res))
    return run


# ----------------------------------------------------------------------------------------- NMS --
NMS_CFGS = [dict(seed=0, N=2000, pre=-1, post=1000, thr=0.7), dict(seed=2, N=1000, pre=600, post=300, thr=0.5),
            dict(seed=4, N=2500, pre=2000, post=2000, thr=0.7, mode="all_overlap")]


def case_nms(i):
    c = NMS_CFGS[i]

    def run(runner):
        dets = np.stack([synth.nms_dets(c["seed"] + k, c["N"], mode=c.get("mode", "clustered")) for k in range(2)])
        if runner == "ref":
            out, score = _ref("nms", "_contrib_NMS", rpn_pre_nms_top_n=c["pre"], rpn_post_nms_top_n=c["post"],
                              threshold=c["thr"], output_score=True).forward([dets], ctx="gpu")
        elif runner == "oracle":
            out, score = _orc().nms(dets, c["pre"], c["post"], c["thr"])[:2]
        else:
            out, score = [_n(t) for t in _ops().nms(_t(dets), c["pre"], c["post"], c["thr"])]
        return {"output": out, "score": score.reshape(out.shape[0], -1, 1)}
    return run


# --------------------------------------------------------------------------------- Proposal_v3 --
PV3_CFGS = [dict(stride=16, H=50, W=84, pre=2000, post=1000, ms=0, train=False),
            dict(stride=32, H=25, W=42, pre=2000, post=1000, ms=16, train=True),
            dict(stride=8, H=100, W=167, pre=2000, post=2000, ms=0, train=False),
            dict(stride=64, H=13, W=21, pre=2000, post=2000, ms=0, train=False),
            dict(stride=4, H=200, W=334, pre=2000, post=2000, ms=8, train=False),
            # iou_loss = true (IoUPredKernel, proposal_v3.cu:163-205; deltas are corner offsets in pixels)
            dict(stride=16, H=50, W=84, pre=2000, post=1000, ms=0, train=False, iou=True),
            dict(stride=32, H=25, W=42, pre=1000, post=300, ms=16, train=True, iou=True)]


def case_proposal_v3(i):
    c = PV3_CFGS[i]

    def run(runner):
        cls, bb, info = synth.rpn_outputs(i, 2, 3, c["H"], c["W"], c["stride"])
        iou = bool(c.get("iou"))
        if iou:
            bb = (bb * 24).astype(np.float32)
        if runner == "ref":
            out, score = _ref("proposal_v3", "_contrib_Proposal_v3", rpn_pre_nms_top_n=c["pre"],
                              rpn_post_nms_top_n=c["post"], threshold=0.7, rpn_min_size=c["ms"], scales=(8,),
                              ratios=(0.5, 1, 2), feature_stride=c["stride"], output_score=True,
                              is_train=c["train"], iou_loss=iou).forward([cls, bb, info], ctx="gpu")
        elif runner == "oracle":
            out, score = _orc().proposal_v3(cls, bb, info, c["pre"], c["post"], 0.7, c["ms"], (8,),
                                            (0.5, 1, 2), c["stride"], c["train"], iou_loss=iou)
        else:
            out, score = [_n(t) for t in _ops().proposal_v3(_t(cls), _t(bb), _t(info), c["pre"], c["post"], 0.7,
                                                             c["ms"], (8,), (0.5, 1, 2), c["stride"], c["train"],
                                                             iou_loss=iou)]
        return {"output": out, "score": score.reshape(out.shape[0], -1, 1)}
    return run


# ---------------------------------------------------------------------------------- DecodeBBox --
def case_decode_bbox(class_agnostic, xyxy, seed=5):
    def run(runner):
        rs = np.random.RandomState(seed)
        rois = synth.random_rois(seed, 2, 300)
        pred = (rs.standard_normal((2, 300, 8 if class_agnostic else 324)) * 0.5).astype(np.float32)
        info = np.array([[800, 1333, 1.5], [768, 1280, 1.2]], np.float32)
        if runner == "ref":
            out = _ref("decodebbox", "_contrib_DecodeBBox", class_agnostic=class_agnostic,
                       bbox_decode_type="xyxy" if xyxy else "xywh", bbox_mean=(0, 0, 0, 0),
                       bbox_std=(0.1, 0.1, 0.2, 0.2)).forward([rois, pred, info])[0]
        elif runner == "oracle":
            out = _orc().decode_bbox(rois, pred, info, class_agnostic=class_agnostic, xyxy=xyxy)
        else:
            out = _n(_ops().decode_bbox(_t(rois), _t(pred), _t(info), class_agnostic=class_agnostic,
                                        bbox_decode_type="xyxy" if xyxy else "xywh"))
        return {"output": out}
    return run


# kind: "exact" -> SHA-256 digests; ("close", tol) -> arrays stored, elementwise |a-b| <= tol
CASES = {}


def _add(name, fn, kind="exact", oracle_exact=True, hip_close=None):
    """hip_close: {output: tol} -- outputs the HIP kernel reproduces to |err| <= tol * max(1, |ref|)
    instead of bit for bit (the oracle still matches them exactly)."""
    CASES[name] = dict(run=fn, kind=kind, oracle_exact=oracle_exact, hip_close=hip_close or {})


for _l in range(4):
    _add("roi_align_v2_P%d_c16" % (_l + 2), case_roi_align_level(_l))
_add("roi_align_v2_P4_14x14_c8", case_roi_align_level(2, channels=8, pooled=(14, 14), num=128, seed=7))
_add("roi_align_v2_C4_c32", case_roi_align_c4())
_add("fpn_roi_align_c8", case_roi_align_fpn(8))
_add("fpn_roi_align_c256_full", case_roi_align_fpn(256))          # BASELINE configs[1] at full size
_add("fpn_roi_align_14x14_c16", case_roi_align_fpn(16, pooled=(14, 14), num=128, seed=7))
for _l in (1, 3):
    # ref == oracle bit for bit (same accumulation order); the HIP scatter sums in another order
    _add("roi_align_v2_bwd_P%d" % (_l + 2), case_roi_align_bwd(_l), kind=("close", 1e-4))
_add("roi_pool_v1_fwd_cpu", case_roi_pool_fwd("cpu"))
_add("roi_pool_v1_fwd_gpu", case_roi_pool_fwd("gpu"))
_add("roi_pool_v1_bwd_gpu", case_roi_pool_bwd(), kind=("close", 1e-4))
for _i in range(len(ANCHOR_CFGS)):
    _add("gen_anchor_%d_cpu" % _i, case_gen_anchor(_i, "cpu"))
    _add("gen_anchor_%d_gpu" % _i, case_gen_anchor(_i, "gpu"))
# bbox_target holds log(gt_w / ex_w): glibc's logf on the host (and which of its ifunc variants runs
# depends on the CPU) against the device logf -- 1e-6 relative; everything else, including the
# libc rand() / random_shuffle replay that picks the rows, is bit-exact
_PT_CLOSE = {"bbox_target": 2e-6}
for _i in range(len(PT_CFGS)):
    _add("proposal_target_%d" % _i, case_proposal_target(_i), hip_close=_PT_CLOSE)
_add("proposal_target_0_agnostic", case_proposal_target(0, class_agnostic=True), hip_close=_PT_CLOSE)
_add("proposal_target_1_third_call", case_proposal_target(1, calls=3), hip_close=_PT_CLOSE)
for _i in range(len(PT2_CFGS)):
    _add("proposal_target_v2_%d" % _i, case_proposal_target_v2(_i), hip_close=_PT_CLOSE)
for _i in range(len(PMT_CFGS)):
    _add("proposal_mask_target_%d" % _i, case_proposal_mask_target(_i), hip_close=_PT_CLOSE)
for _i in range(len(NMS_CFGS)):
    _add("nms_%d" % _i, case_nms(_i))
for _i in range(len(PV3_CFGS)):
    # CUDA exp(float) is emulated with glibc expf, the oracle / HIP kernel use the correctly rounded
    # (float)exp((double)x): coordinates may differ by one ulp of a <=2^11 pixel value
    if PV3_CFGS[_i].get("iou"):  # no exp in IoUPredKernel: bit exact
        _add("proposal_v3_%d_iou_loss" % _i, case_proposal_v3(_i))
    else:
        _add("proposal_v3_%d" % _i, case_proposal_v3(_i), kind=("close", 2.5e-4), oracle_exact=False)
for _ca in (True, False):
    for _xy in (False, True):
        _add("decode_bbox_%s_%s" % ("agn" if _ca else "cls", "xyxy" if _xy else "xywh"), case_decode_bbox(_ca, _xy))


def run_case(name, runner):
    return CASES[name]["run"](runner)
