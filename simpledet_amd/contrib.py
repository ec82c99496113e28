"""The reference's operator names for the PyTorch-ROCm harness (autograd enabled).

Each function mirrors the symbol constructor of the same name in the reference graph
(`mx.sym.contrib.ROIAlign_v2`, `mx.sym.ROIPooling_v1`, `mx.sym.ProposalTarget`,
`mx.sym.contrib.GenAnchor`, `mx.sym.contrib.NMS`, `mx.sym.contrib.DeformableConvolution`, mxnext's
`X.roi_align`): same argument names and meaning, same visible outputs.  Forward and backward both
go through the HIP C ABI (simpledet_amd.ops); there is no CPU path.
"""
import os

import torch

from . import ops


class _ROIAlignV2(torch.autograd.Function):
    @staticmethod
    def forward(ctx, data, rois, pooled_size, spatial_scale):
        out, mx, my = ops.roi_align_v2_forward(data, rois, pooled_size, spatial_scale)
        ctx.save_for_backward(rois, mx, my)
        ctx.data_shape, ctx.scale = tuple(data.shape), spatial_scale
        return out

    @staticmethod
    def backward(ctx, dy):
        rois, mx, my = ctx.saved_tensors
        dx, _ = ops.roi_align_v2_backward(dy.contiguous(), rois, mx, my, ctx.data_shape, ctx.scale,
                                          req_rois="null")
        return dx, None, None, None


def ROIAlign_v2(data, rois, pooled_size, spatial_scale):
    """mx.sym.contrib.ROIAlign_v2: data (B,C,H,W), rois (B,R,4) -> (B,R,C,ph,pw) (1 visible
    output; maxidx_x / maxidx_y are kept for the backward, roi_align_v2.cc:175-186)."""
    return _ROIAlignV2.apply(data, rois, pooled_size, spatial_scale)


def roi_align(data, rois, out_size, stride):
    """mxnext X.roi_align(feat, rois, out_size, stride) (symbol/builder.py:885-891)."""
    return ROIAlign_v2(data, rois, (out_size, out_size), 1.0 / stride)


class _FPNRoIAlign(torch.autograd.Function):
    @staticmethod
    def forward(ctx, rois, rcnn_stride, pooled_size, scale0, lvl0, *feats):
        ph, pw = (pooled_size, pooled_size) if isinstance(pooled_size, int) else tuple(pooled_size)
        ctx.packed = (ph, pw) in ((7, 7), (14, 14)) and rois.shape[1] <= 8192
        if ctx.packed:  # one-byte arg-max between forward and backward
            out, am = ops.fpn_roi_align_forward_packed(list(feats), rois, rcnn_stride, (ph, pw),
                                                       scale0, lvl0)
            ctx.save_for_backward(rois, am[0], am[1])
        else:
            out, mx, my = ops.fpn_roi_align_forward(list(feats), rois, rcnn_stride, (ph, pw), scale0,
                                                    lvl0)
            ctx.save_for_backward(rois, mx, my)
        ctx.meta = ([tuple(f.shape) for f in feats], list(rcnn_stride), scale0, lvl0)
        return out

    @staticmethod
    def backward(ctx, dy):
        shapes, strides, scale0, lvl0 = ctx.meta
        if ctx.packed:
            rois, am, co = ctx.saved_tensors
            d = ops.fpn_roi_align_backward_packed(dy.contiguous(), rois, (am, co), shapes, strides,
                                                  scale0, lvl0)
        else:
            rois, mx, my = ctx.saved_tensors
            d = ops.fpn_roi_align_backward(dy.contiguous(), rois, mx, my, shapes, strides, scale0,
                                           lvl0)
        return (None, None, None, None, None) + tuple(d)


def fpn_roi_align(feats, rois, rcnn_stride, pooled_size=(7, 7), roi_canonical_scale=224,
                  roi_canonical_level=4):
    """FPNRoiAlign.get_roi_feature (models/FPN/builder.py:567-610) as ONE op:
    fpn_roi_assign -> per-level ROIAlign_v2 -> add_n.  feats: list of (B,C,H_l,W_l)."""
    return _FPNRoIAlign.apply(rois, tuple(rcnn_stride), pooled_size, roi_canonical_scale,
                              roi_canonical_level, *feats)


class _ROIPoolingV1(torch.autograd.Function):
    @staticmethod
    def forward(ctx, data, rois, pooled_size, spatial_scale):
        out, idx = ops.roi_pool_v1_forward(data, rois, pooled_size, spatial_scale)
        ctx.save_for_backward(rois, idx)
        ctx.data_shape, ctx.scale = tuple(data.shape), spatial_scale
        return out

    @staticmethod
    def backward(ctx, dy):
        rois, idx = ctx.saved_tensors
        dx, _ = ops.roi_pool_v1_backward(dy.contiguous(), rois, idx, ctx.data_shape, ctx.scale,
                                         req_rois="null")
        return dx, None, None, None


def ROIPooling_v1(data, rois, pooled_size, spatial_scale):
    """mx.sym.ROIPooling_v1: data (B,C,H,W), rois (K,5) -> (K,C,ph,pw)."""
    return _ROIPoolingV1.apply(data, rois, pooled_size, spatial_scale)


class _DeformConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, data, offset, weight, bias, pad, stride, dilate, dg, num_group, will_backward):
        ctx.save_for_backward(data, offset, weight)
        ctx.cfg = (pad, stride, dilate, dg, num_group, bias is not None)
        # the col matrix stays alive for the backward (which then skips its own im2col): 620 MB per
        # layer at (16,256,50,84) -- only when a backward can follow (will_backward: grad mode on and
        # some input requires a gradient, decided by the caller: in here grad mode is always off).
        # Inference and SIMPLEDET_AMD_DCN_CACHE_COL=0 (process-wide) take the col-free fused kernel; the
        # backward then recomputes col.
        keep = will_backward and os.environ.get("SIMPLEDET_AMD_DCN_CACHE_COL", "1") != "0"
        if keep:
            y, ctx.fwd_ws = ops.deform_conv_forward(data, offset, weight, pad, stride, dilate, dg, keep_col=True,
                                                    bias=bias, num_group=num_group)
        else:
            y, ctx.fwd_ws = ops.deform_conv_forward(data, offset, weight, pad, stride, dilate, dg, bias=bias,
                                                    num_group=num_group), None
        return y

    @staticmethod
    def backward(ctx, dy):
        data, offset, weight = ctx.saved_tensors
        pad, stride, dilate, dg, num_group, has_bias = ctx.cfg
        need = ctx.needs_input_grad
        req = tuple("write" if n else "null" for n in need[:3])
        if has_bias:
            req = req + ("write" if need[3] else "null",)
        g = ops.deform_conv_backward(dy.contiguous(), data, offset, weight, pad, stride, dilate, dg, req=req,
                                     fwd_ws=ctx.fwd_ws, num_group=num_group, bias=has_bias)
        ctx.fwd_ws = None
        return (g[0] if need[0] else None, g[1] if need[1] else None, g[2] if need[2] else None,
                g[3] if has_bias and need[3] else None, None, None, None, None, None, None)


def DeformableConvolution(data, offset, weight, *, bias=None, kernel=(3, 3), stride=(1, 1), dilate=(1, 1),
                          pad=(0, 0), num_filter=None, num_group=1, num_deformable_group=1,
                          no_bias=False):
    """mx.sym.contrib.DeformableConvolution with upstream's parameter set (no_bias defaults to False
    there): models/dcn/builder.py:14-17 (no_bias=True, 4 deformable groups), models/RepPoints/builder.py:
    215-245 (bias), models/sepc/sepc_dconv.py:12-16 (num_group / bias passed through).  Everything behind the
    three tensors is keyword-only: a positional `kernel` must not land in `bias`."""
    if no_bias:
        if bias is not None:
            raise ValueError("no_bias=True but a bias was given")
    elif bias is None:
        raise ValueError("no_bias=False needs a bias (F,)")
    k, s, d, p = [v if isinstance(v, (tuple, list)) else (v, v) for v in (kernel, stride, dilate, pad)]
    if tuple(weight.shape[2:]) != tuple(k) or (num_filter is not None and weight.shape[0] != num_filter):
        raise ValueError("weight shape %s does not match kernel/num_filter" % (tuple(weight.shape),))
    if s[0] != s[1] or d[0] != d[1] or p[0] != p[1]:
        raise ValueError("square stride/dilate/pad only")
    will_backward = torch.is_grad_enabled() and any(t is not None and t.requires_grad
                                                    for t in (data, offset, weight, bias))
    return _DeformConv.apply(data, offset, weight, bias, p[0], s[0], d[0], num_deformable_group, int(num_group),
                             will_backward)


def ProposalTarget(rois, gt_boxes, num_classes, batch_images, image_rois, fg_thresh, bg_thresh_hi,
                   bg_thresh_lo, fg_fraction=0.25, proposal_without_gt=False, class_agnostic=False,
                   output_iou=False, bbox_mean=(0., 0., 0., 0.), bbox_std=(.1, .1, .2, .2),
                   bbox_weight=(1., 1., 1., 1.), rng_state=None):
    """mx.sym.ProposalTarget: 4 visible outputs (5 with output_iou, proposal_target-inl.h:297-303);
    no gradient flows to rois / gt_boxes (:272-276)."""
    with torch.no_grad():
        res = ops.proposal_target(rois, gt_boxes, num_classes, batch_images, image_rois, fg_fraction,
                                  fg_thresh, bg_thresh_hi, bg_thresh_lo, proposal_without_gt,
                                  class_agnostic, bbox_mean, bbox_std, bbox_weight, rng_state)
    return res if output_iou else res[:4]


def GenAnchor(cls_prob, feature_stride=16, scales=(4.0, 8.0, 16.0, 32.0), ratios=(0.5, 1.0, 2.0)):
    """mx.sym.contrib.GenAnchor: only the (H, W) of cls_prob is used."""
    return ops.gen_anchor(cls_prob.shape[2], cls_prob.shape[3], feature_stride, scales, ratios,
                          device=cls_prob.device)


def NMS(rois, rpn_pre_nms_top_n=6000, rpn_post_nms_top_n=300, threshold=0.7, output_score=False,
        already_sorted=False):
    """mx.sym.contrib.NMS: output (B,post,4) [, score (B,post,1) with output_score]."""
    with torch.no_grad():
        out, score = ops.nms(rois, rpn_pre_nms_top_n, rpn_post_nms_top_n, threshold, already_sorted)
    return (out, score) if output_score else out


def Proposal_v3(cls_prob, bbox_pred, im_info, rpn_pre_nms_top_n=6000, rpn_post_nms_top_n=300,
                threshold=0.7, rpn_min_size=16, scales=(4., 8., 16., 32.), ratios=(0.5, 1., 2.),
                feature_stride=16, output_score=False, iou_loss=False, is_train=False):
    """mx.sym.contrib.Proposal_v3 as models/FPN/builder.py:275-287 calls it."""
    if iou_loss:
        raise ValueError("Proposal_v3: iou_loss=True is not supported")
    with torch.no_grad():
        out, score = ops.proposal_v3(cls_prob, bbox_pred, im_info, rpn_pre_nms_top_n,
                                     rpn_post_nms_top_n, threshold, rpn_min_size, scales, ratios,
                                     feature_stride, is_train)
    return (out, score) if output_score else out


def get_top_proposal(bbox, score, top_n):
    """mxnext.tvm.get_top_proposal / models/FPN/get_top_proposal.py."""
    with torch.no_grad():
        return ops.get_top_proposal(bbox, score, top_n)


def DecodeBBox(rois, bbox_pred, im_info, bbox_mean=(0., 0., 0., 0.), bbox_std=(.1, .1, .2, .2),
               class_agnostic=True, bbox_decode_type="xywh"):
    """mx.sym.contrib.DecodeBBox / X.decode_bbox (symbol/builder.py:384-392)."""
    with torch.no_grad():
        return ops.decode_bbox(rois, bbox_pred, im_info, bbox_mean, bbox_std, class_agnostic,
                               bbox_decode_type)
