"""numpy binding of oracle/liboracle.so -- TEST INFRASTRUCTURE ONLY.

May be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg, never by
anything under simpledet_amd/.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "liboracle.so")
_F = ctypes.POINTER(ctypes.c_float)
_cdll = None


def build(force=False):
    if force or not os.path.exists(_LIB):
        subprocess.check_call(["make", "-s", "-C", _HERE] + (["-B"] if force else []))
    return _LIB


def cdll():
    global _cdll
    if _cdll is None:
        build()
        _cdll = ctypes.CDLL(_LIB)
    return _cdll


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(ctypes.c_void_p)


def _i(vals):
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


def roi_align_v2_fwd(data, rois, pooled_size, spatial_scale, nthreads=1):
    data, pd = _f(data)
    rois, pr = _f(rois)
    B, C, H, W = data.shape
    R = rois.shape[1]
    ph, pw = pooled_size
    out = np.empty((B, R, C, ph, pw), np.float32)
    ax = np.empty_like(out)
    ay = np.empty_like(out)
    cdll().orc_roi_align_v2_fwd(pd, pr, out.ctypes, ax.ctypes, ay.ctypes, B, C, H, W, R, ph, pw,
                                ctypes.c_float(spatial_scale), int(nthreads))
    return out, ax, ay


def roi_align_v2_bwd(dy, ax, ay, data_shape, req=1, dx=None, nthreads=1):
    dy, pdy = _f(dy)
    ax, pax = _f(ax)
    ay, pay = _f(ay)
    B, C, H, W = data_shape
    _, R, _, ph, pw = dy.shape
    if dx is None:
        dx = np.zeros((B, C, H, W), np.float32)
    cdll().orc_roi_align_v2_bwd(pdy, pax, pay, dx.ctypes, B, C, H, W, R, ph, pw, int(req),
                                int(nthreads))
    return dx


def roi_align_v2_bwd_cpu_gather(dy, rois, ax, ay, data_shape, spatial_scale, req=1):
    dy, pdy = _f(dy)
    rois, pr = _f(rois)
    ax, pax = _f(ax)
    ay, pay = _f(ay)
    B, C, H, W = data_shape
    _, R, _, ph, pw = dy.shape
    dx = np.zeros((B, C, H, W), np.float32)
    cdll().orc_roi_align_v2_bwd_cpu_gather(pdy, pr, pax, pay, dx.ctypes, B, C, H, W, R, ph, pw,
                                           ctypes.c_float(spatial_scale), int(req))
    return dx


def fpn_roi_assign(rois, strides, canonical_scale=224, canonical_level=4):
    rois, pr = _f(rois)
    n = int(np.prod(rois.shape[:-1]))
    level = np.empty(n, np.int32)
    per = np.empty((len(strides),) + rois.shape, np.float32)
    cdll().orc_fpn_roi_assign(pr, n, _i(strides), len(strides), ctypes.c_float(canonical_scale),
                              ctypes.c_float(canonical_level), level.ctypes, per.ctypes)
    return per, level.reshape(rois.shape[:-1])


def _pp(arrs):
    return (ctypes.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])


def fpn_roi_align_fwd(feats, rois, strides, pooled_size, canonical_scale=224, canonical_level=4,
                      nthreads=1):
    feats = [np.ascontiguousarray(f, np.float32) for f in feats]
    rois, pr = _f(rois)
    B, C = feats[0].shape[:2]
    R = rois.shape[1]
    ph, pw = pooled_size
    out = np.empty((B, R, C, ph, pw), np.float32)
    ax = np.empty_like(out)
    ay = np.empty_like(out)
    cdll().orc_fpn_roi_align_fwd(_pp(feats), _i([f.shape[2] for f in feats]),
                                 _i([f.shape[3] for f in feats]), _i(strides), len(feats), pr,
                                 out.ctypes, ax.ctypes, ay.ctypes, B, C, R, ph, pw,
                                 ctypes.c_float(canonical_scale), ctypes.c_float(canonical_level),
                                 int(nthreads))
    return out, ax, ay


def fpn_roi_align_bwd(dy, rois, ax, ay, feat_shapes, strides, canonical_scale=224,
                      canonical_level=4, req=1, dfeats=None, nthreads=1):
    dy, pdy = _f(dy)
    rois, pr = _f(rois)
    ax, pax = _f(ax)
    ay, pay = _f(ay)
    B, R, C, ph, pw = dy.shape
    if dfeats is None:
        dfeats = [np.zeros(tuple(s), np.float32) for s in feat_shapes]
    cdll().orc_fpn_roi_align_bwd(pdy, pr, pax, pay, _pp(dfeats), _i([f.shape[2] for f in dfeats]),
                                 _i([f.shape[3] for f in dfeats]), _i(strides), len(dfeats), B, C,
                                 R, ph, pw, ctypes.c_float(canonical_scale),
                                 ctypes.c_float(canonical_level), int(req), int(nthreads))
    return dfeats


# ------------------------------------------------------------------------------------------------
# ROIPooling_v1
# ------------------------------------------------------------------------------------------------
def roi_pool_v1_fwd(data, rois, pooled_size, spatial_scale):
    data, pd = _f(data)
    rois, pr = _f(rois)
    B, C, H, W = data.shape
    K = rois.shape[0]
    ph, pw = pooled_size
    out = np.empty((K, C, ph, pw), np.float32)
    idx = np.empty_like(out)
    cdll().orc_roi_pool_v1_fwd(pd, pr, out.ctypes, idx.ctypes, B, C, H, W, K, ph, pw,
                               ctypes.c_float(spatial_scale))
    return out, idx


def roi_pool_v1_bwd(dy, rois, maxidx, data_shape, spatial_scale, req=1, dx=None, gather=False):
    dy, pdy = _f(dy)
    rois, pr = _f(rois)
    maxidx, pm = _f(maxidx)
    B, C, H, W = data_shape
    K, _, ph, pw = dy.shape
    if dx is None:
        dx = np.zeros((B, C, H, W), np.float32)
    fn = cdll().orc_roi_pool_v1_bwd_cpu_gather if gather else cdll().orc_roi_pool_v1_bwd
    fn(pdy, pr, pm, dx.ctypes, B, C, H, W, K, ph, pw, ctypes.c_float(spatial_scale), int(req))
    return dx


# ------------------------------------------------------------------------------------------------
# GenAnchor
# ------------------------------------------------------------------------------------------------
def _d(vals):
    return (ctypes.c_double * len(vals))(*[float(v) for v in vals])


def gen_base_anchors(stride, scales, ratios):
    base = np.empty((len(ratios) * len(scales), 4), np.float64)
    cdll().orc_gen_base_anchors(int(stride), _d(scales), len(scales), _d(ratios), len(ratios),
                                base.ctypes)
    return base


def gen_anchor(H, W, stride, scales, ratios):
    out = np.empty((H * W * len(scales) * len(ratios), 4), np.float32)
    cdll().orc_gen_anchor(out.ctypes, int(H), int(W), int(stride), _d(scales), len(scales),
                          _d(ratios), len(ratios))
    return out


# ------------------------------------------------------------------------------------------------
# ProposalTarget
# ------------------------------------------------------------------------------------------------
class ProposalTargetParam(ctypes.Structure):
    _fields_ = [("num_classes", ctypes.c_int), ("batch_images", ctypes.c_int),
                ("image_rois", ctypes.c_int), ("fg_fraction", ctypes.c_float),
                ("fg_thresh", ctypes.c_float), ("bg_thresh_hi", ctypes.c_float),
                ("bg_thresh_lo", ctypes.c_float), ("proposal_without_gt", ctypes.c_int),
                ("class_agnostic", ctypes.c_int), ("bbox_mean", ctypes.c_float * 4),
                ("bbox_std", ctypes.c_float * 4), ("bbox_weight", ctypes.c_float * 4)]


class GlibcRand(ctypes.Structure):
    """glibc TYPE_3 rand() state (orc_glibc_rand)."""
    _fields_ = [("r", ctypes.c_int32 * 34), ("f", ctypes.c_int32), ("b", ctypes.c_int32)]

    def __init__(self, seed=1):
        super().__init__()
        cdll().orc_glibc_srand(ctypes.byref(self), ctypes.c_uint(seed))

    def next(self):
        return int(cdll().orc_glibc_rand_next(ctypes.byref(self)))

    def state_words(self):
        """(ring of 31 words, f, b) -- the layout sd_proposal_target's rng_state uses."""
        return np.array(list(self.r)[:31] + [self.f, self.b], np.int32)


def make_pt_param(num_classes, batch_images, image_rois, fg_fraction=0.25, fg_thresh=0.5,
                  bg_thresh_hi=0.5, bg_thresh_lo=0.0, proposal_without_gt=False,
                  class_agnostic=False, bbox_mean=(0, 0, 0, 0), bbox_std=(0.1, 0.1, 0.2, 0.2),
                  bbox_weight=(1, 1, 1, 1)):
    p = ProposalTargetParam()
    p.num_classes, p.batch_images, p.image_rois = int(num_classes), int(batch_images), int(image_rois)
    p.fg_fraction, p.fg_thresh = fg_fraction, fg_thresh
    p.bg_thresh_hi, p.bg_thresh_lo = bg_thresh_hi, bg_thresh_lo
    p.proposal_without_gt, p.class_agnostic = int(proposal_without_gt), int(class_agnostic)
    for i in range(4):
        p.bbox_mean[i], p.bbox_std[i], p.bbox_weight[i] = bbox_mean[i], bbox_std[i], bbox_weight[i]
    return p


def proposal_target(rois, gt_boxes, param, rng=None, use_libc=False, valid_ranges=None,
                    filter_scales=False):
    """Returns (rois_out, label, bbox_target, bbox_weight, match_gt_iou, kept_index, rc).
    valid_ranges (B,2) selects ProposalTarget_v2."""
    rois, pr = _f(rois)
    gt, pg = _f(gt_boxes)
    B, N, _ = rois.shape
    M = gt.shape[1]
    S, K4 = param.image_rois, 4 * param.num_classes
    ro = np.empty((B, S, 4), np.float32)
    lb = np.empty((B, S), np.float32)
    bt = np.empty((B, S, K4), np.float32)
    bw = np.empty((B, S, K4), np.float32)
    iou = np.empty((B, S), np.float32)
    kept = np.empty((B, S), np.int32)
    if valid_ranges is not None:
        vr, pv = _f(valid_ranges)
        if rng is None:
            rng = GlibcRand(1)
        rc = cdll().orc_proposal_target_v2(pr, pg, pv, int(filter_scales), N, M, ctypes.byref(param),
                                           ctypes.byref(rng), ro.ctypes, lb.ctypes, bt.ctypes,
                                           bw.ctypes, iou.ctypes, kept.ctypes)
    elif use_libc:
        rc = cdll().orc_proposal_target_libc(pr, pg, N, M, ctypes.byref(param), ro.ctypes,
                                             lb.ctypes, bt.ctypes, bw.ctypes, iou.ctypes,
                                             kept.ctypes)
    else:
        if rng is None:
            rng = GlibcRand(1)
        rc = cdll().orc_proposal_target(pr, pg, N, M, ctypes.byref(param), ctypes.byref(rng),
                                        ro.ctypes, lb.ctypes, bt.ctypes, bw.ctypes, iou.ctypes,
                                        kept.ctypes)
    return ro, lb, bt, bw, iou, kept, rc


def proposal_mask_target(rois, gt_boxes, gt_polys, param, mask_size=28, rng=None, valid_ranges=None,
                         filter_scales=False, output_ratio=False):
    """ProposalMaskTarget: the six outputs + kept_index; with output_ratio=True the mask ratio
    (B, FG) of proposal_mask_target.cc:20-152 is appended after kept_index."""
    rois, pr = _f(rois)
    gt, pg = _f(gt_boxes)
    polys, pp = _f(gt_polys)
    B, N, _ = rois.shape
    M, L = gt.shape[1], polys.shape[2]
    S, K4 = param.image_rois, 4 * param.num_classes
    FG = int(np.float32(S) * np.float32(param.fg_fraction))
    ro = np.empty((B, S, 4), np.float32)
    lb = np.empty((B, S), np.float32)
    bt = np.empty((B, S, K4), np.float32)
    bw = np.empty((B, S, K4), np.float32)
    iou = np.empty((B, S), np.float32)
    kept = np.empty((B, S), np.int32)
    mask = np.empty((B, FG, mask_size, mask_size), np.float32)
    pv = None
    if valid_ranges is not None:
        vr, pv = _f(valid_ranges)
    if rng is None:
        rng = GlibcRand(1)
    if output_ratio:
        ratio = np.empty((B, FG), np.float32)
        cdll().orc_proposal_mask_target_ratio(pr, pg, pp, pv, int(filter_scales), N, M, L, int(mask_size),
                                              ctypes.byref(param), ctypes.byref(rng), ro.ctypes, lb.ctypes,
                                              bt.ctypes, bw.ctypes, iou.ctypes, kept.ctypes, mask.ctypes,
                                              ratio.ctypes)
        return ro, lb, bt, bw, iou, mask, kept, ratio
    cdll().orc_proposal_mask_target(pr, pg, pp, pv, int(filter_scales), N, M, L, int(mask_size),
                                    ctypes.byref(param), ctypes.byref(rng), ro.ctypes, lb.ctypes,
                                    bt.ctypes, bw.ctypes, iou.ctypes, kept.ctypes, mask.ctypes)
    return ro, lb, bt, bw, iou, mask, kept


def poly2mask(roi, poly, mask_size=28):
    roi, pr = _f(roi)
    poly, pp = _f(poly)
    m = np.empty((mask_size, mask_size), np.float32)
    cdll().orc_poly2mask(pr, pp, int(mask_size), m.ctypes)
    return m


def poly2mask_ratio(roi, poly, mask_size=28):
    roi, pr = _f(roi)
    poly, pp = _f(poly)
    m = np.empty((mask_size, mask_size), np.float32)
    r = ctypes.c_double(0)
    cdll().orc_poly2mask_ratio(pr, pp, int(mask_size), m.ctypes, ctypes.byref(r))
    return m, r.value


def std_random_shuffle(a):
    a = np.ascontiguousarray(a, np.uint32).copy()
    cdll().orc_std_random_shuffle(a.ctypes, len(a))
    return a


def libc_srand(seed):
    cdll().orc_libc_srand(ctypes.c_uint(seed))


def libc_rand():
    return int(cdll().orc_libc_rand())


# ------------------------------------------------------------------------------------------------
# NMS family
# ------------------------------------------------------------------------------------------------
def nms(dets, pre_nms_top_n, post_nms_top_n, threshold, already_sorted=False):
    """_contrib_NMS (GPU path).  dets (B,N,5) -> out (B,post,4), score (B,post,1), keep (B,post)."""
    dets, pd = _f(dets)
    B, N, _ = dets.shape
    pre = pre_nms_top_n if pre_nms_top_n > 0 else N
    pre = min(pre, N)
    post = min(post_nms_top_n, pre)
    out = np.zeros((B, post, 4), np.float32)
    score = np.zeros((B, post, 1), np.float32)
    keep = np.zeros((B, post), np.int32)
    cdll().orc_nms(pd, B, N, int(pre_nms_top_n), int(post_nms_top_n), ctypes.c_float(threshold),
                   int(already_sorted), out.ctypes, score.ctypes, keep.ctypes)
    return out, score, keep


def soft_nms(boxes, sigma=0.5, Nt=0.3, threshold=0.001, method=0):
    boxes = np.array(boxes, dtype=np.float32, order="C", copy=True)
    n = boxes.shape[0]
    inds = np.empty(n, np.int64)
    cdll().orc_soft_nms.restype = ctypes.c_int
    N = cdll().orc_soft_nms(boxes.ctypes, inds.ctypes, n, ctypes.c_float(sigma),
                            ctypes.c_float(Nt), ctypes.c_float(threshold), ctypes.c_uint(method))
    return boxes[:N], inds[:N]


def greedy_nms(dets, thresh):
    dets, pd = _f(dets)
    keep = np.empty(dets.shape[0], np.int64)
    n = cdll().orc_greedy_nms(pd, dets.shape[0], ctypes.c_float(thresh), keep.ctypes)
    return keep[:n]


def bbox_overlaps(boxes, query):
    boxes, pb = _f(boxes)
    query, pq = _f(query)
    ov = np.empty((boxes.shape[0], query.shape[0]), np.float32)
    cdll().orc_bbox_overlaps(pb, boxes.shape[0], pq, query.shape[0], ov.ctypes)
    return ov


# ------------------------------------------------------------------------------------------------
# DeformableConvolution v1 (parity unpinned: restated published algorithm, see deform_conv.c)
# ------------------------------------------------------------------------------------------------
def _out_hw(H, W, kh, kw, pad, stride, dil):
    return ((H + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1,
            (W + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1)


def deform_im2col(x, offset, kernel=(3, 3), pad=1, stride=1, dil=1, dgroup=1):
    """x (C,H,W), offset (dgroup*2*kh*kw,Ho,Wo) -> col (C*kh*kw, Ho*Wo)."""
    x, px = _f(x)
    offset, po = _f(offset)
    C, H, W = x.shape
    kh, kw = kernel
    Ho, Wo = _out_hw(H, W, kh, kw, pad, stride, dil)
    col = np.empty((C * kh * kw, Ho * Wo), np.float32)
    cdll().orc_deform_im2col(px, po, col.ctypes, C, H, W, kh, kw, pad, pad, stride, stride, dil, dil,
                             dgroup, Ho, Wo)
    return col


def deform_col2im(col, offset, x_shape, kernel=(3, 3), pad=1, stride=1, dil=1, dgroup=1):
    col, pc = _f(col)
    offset, po = _f(offset)
    C, H, W = x_shape
    kh, kw = kernel
    Ho, Wo = _out_hw(H, W, kh, kw, pad, stride, dil)
    dx = np.zeros((C, H, W), np.float32)
    cdll().orc_deform_col2im(pc, po, dx.ctypes, C, H, W, kh, kw, pad, pad, stride, stride, dil, dil,
                             dgroup, Ho, Wo)
    return dx


def deform_col2im_coord(col, x, offset, kernel=(3, 3), pad=1, stride=1, dil=1, dgroup=1):
    col, pc = _f(col)
    x, px = _f(x)
    offset, po = _f(offset)
    C, H, W = x.shape
    kh, kw = kernel
    Ho, Wo = _out_hw(H, W, kh, kw, pad, stride, dil)
    doff = np.zeros_like(offset)
    cdll().orc_deform_col2im_coord(pc, px, po, doff.ctypes, C, H, W, kh, kw, pad, pad, stride,
                                   stride, dil, dil, dgroup, Ho, Wo)
    return doff


def deform_conv_fwd(x, offset, weight, pad=1, stride=1, dil=1, dgroup=1):
    """x (N,C,H,W), offset (N,dgroup*2*kh*kw,Ho,Wo), weight (F,C,kh,kw) -> y (N,F,Ho,Wo)."""
    x, px = _f(x)
    offset, po = _f(offset)
    weight, pw = _f(weight)
    N, C, H, W = x.shape
    F, _, kh, kw = weight.shape
    Ho, Wo = _out_hw(H, W, kh, kw, pad, stride, dil)
    y = np.empty((N, F, Ho, Wo), np.float32)
    cdll().orc_deform_conv_fwd(px, po, pw, y.ctypes, N, C, H, W, F, kh, kw, pad, stride, dil, dgroup)
    return y


def deform_convolution_fwd(x, offset, weight, bias=None, pad=1, stride=1, dil=1, dgroup=1, num_group=1):
    """the operator with every parameter: weight (F, C/num_group, kh, kw), bias (F) or None -> y"""
    x, px = _f(x)
    offset, po = _f(offset)
    weight, pw = _f(weight)
    N, C, H, W = x.shape
    F, Cg, kh, kw = weight.shape
    assert Cg * num_group == C and F % num_group == 0
    pb = None
    if bias is not None:
        bias, pb = _f(bias)
    Ho, Wo = _out_hw(H, W, kh, kw, pad, stride, dil)
    y = np.empty((N, F, Ho, Wo), np.float32)
    cdll().orc_deform_convolution_fwd(px, po, pw, pb, y.ctypes, N, C, H, W, F, kh, kw, pad, stride, dil, dgroup,
                                      num_group)
    return y


def deform_convolution_bwd(dy, x, offset, weight, bias=False, pad=1, stride=1, dil=1, dgroup=1, num_group=1):
    """-> d_x, d_offset, d_weight[, d_bias] (all written)"""
    dy, pdy = _f(dy)
    x, px = _f(x)
    offset, po = _f(offset)
    weight, pw = _f(weight)
    N, C, H, W = x.shape
    F, Cg, kh, kw = weight.shape
    assert Cg * num_group == C and F % num_group == 0
    dx, doff, dw = np.zeros_like(x), np.zeros_like(offset), np.zeros_like(weight)
    db = np.zeros(F, np.float32) if bias else None
    cdll().orc_deform_convolution_bwd(pdy, px, po, pw, dx.ctypes, doff.ctypes, dw.ctypes,
                                      db.ctypes if bias else None, N, C, H, W, F, kh, kw, pad, stride, dil,
                                      dgroup, num_group)
    return (dx, doff, dw, db) if bias else (dx, doff, dw)


# ------------------------------------------------------------------------------------------------
# _contrib_Proposal_v3 (GPU path) and get_top_proposal
# ------------------------------------------------------------------------------------------------
def _fa(vals):
    return (ctypes.c_float * len(vals))(*[float(v) for v in vals])


def proposal_v3_anchors(stride, scales, ratios):
    out = np.empty((len(scales) * len(ratios), 4), np.float32)
    cdll().orc_proposal_v3_anchors(int(stride), _fa(scales), len(scales), _fa(ratios), len(ratios),
                                   out.ctypes)
    return out


def proposal_v3(cls_prob, bbox_pred, im_info, pre, post, threshold, min_size, scales, ratios,
                stride, is_train=False, iou_loss=False):
    cls_prob, pc = _f(cls_prob)
    bbox_pred, pb = _f(bbox_pred)
    im_info, pi = _f(im_info)
    B, A2, H, W = cls_prob.shape
    A = A2 // 2
    cdll().orc_proposal_v3_post.restype = ctypes.c_int
    peff = cdll().orc_proposal_v3_post(A * H * W, int(pre), int(post), int(is_train))
    out = np.empty((B, peff, 4), np.float32)
    score = np.empty((B, peff, 1), np.float32)
    fn = cdll().orc_proposal_v3_iou if iou_loss else cdll().orc_proposal_v3
    fn(pc, pb, pi, B, A, H, W, int(pre), int(post), ctypes.c_float(threshold),
       int(min_size), _fa(scales), len(scales), _fa(ratios), len(ratios),
       int(stride), int(is_train), out.ctypes, score.ctypes)
    return out, score


def get_top_proposal(bbox, score, top_n):
    bbox, pb = _f(bbox)
    score, ps = _f(score)
    B, N = bbox.shape[:2]
    ob = np.empty((B, top_n, 4), np.float32)
    os_ = np.empty((B, top_n, 1), np.float32)
    cdll().orc_get_top_proposal(pb, ps, B, N, int(top_n), ob.ctypes, os_.ctypes)
    return ob, os_


# ------------------------------------------------------------------------------------------------
# _contrib_DecodeBBox + test-time per-class filter
# ------------------------------------------------------------------------------------------------
def decode_bbox(rois, bbox_pred, im_info, bbox_mean=(0, 0, 0, 0), bbox_std=(0.1, 0.1, 0.2, 0.2),
                class_agnostic=True, xyxy=False):
    rois, pr = _f(rois)
    bbox_pred, pb = _f(bbox_pred)
    im_info, pi = _f(im_info)
    B, R, _ = rois.shape
    K = bbox_pred.shape[2] // 4
    out = np.empty((B, R, 4 if class_agnostic else 4 * K), np.float32)
    cdll().orc_decode_bbox(pr, pb, pi, out.ctypes, B, R, K, _fa(bbox_mean), _fa(bbox_std),
                           int(class_agnostic), int(xyxy))
    return out


def det_filter(bbox, cls_score, min_det_score):
    bbox, pb = _f(bbox)
    cls_score, ps = _f(cls_score)
    B, R, K = cls_score.shape
    Kb = bbox.shape[2] // 4
    dets = np.zeros((B * K, R, 5), np.float32)
    counts = np.zeros(B * K, np.int32)
    cdll().orc_det_filter(pb, ps, B, R, K, Kb, ctypes.c_float(min_det_score), dets.ctypes,
                          counts.ctypes)
    return dets, counts
