"""Torch-tensor harness over the C ABI (include/simpledet_ops.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; every function passes raw
device pointers + the current stream to libsimpledet_ops_hip.so.  Nothing in this module computes
on the CPU and nothing falls back: non-CUDA tensors raise.

Function names/arguments mirror the reference operators (operator_cxx/, see each docstring).
"""
import ctypes

import torch

from ._lib import lib

REQ = {"null": 0, "write": 1, "add": 3}


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t, name, dtype=torch.float32, ndim=None):
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if not t.is_cuda:
        raise RuntimeError("%s must live on the GPU (simpledet_amd has no CPU path)" % name)
    if t.dtype != dtype:
        raise TypeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    if ndim is not None and t.dim() != ndim:
        raise ValueError("%s should be a %dD tensor, got shape %s" % (name, ndim, tuple(t.shape)))
    if not t.is_contiguous():
        raise ValueError("%s must be contiguous" % name)
    return t


def _pair(v):
    if isinstance(v, (tuple, list)):
        if len(v) != 2:
            raise ValueError("pooled_size must have 2 entries (h, w)")
        return int(v[0]), int(v[1])
    return int(v), int(v)


def _iarr(vals):
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


def _parr(tensors):
    return (ctypes.c_void_p * len(tensors))(*[t.data_ptr() for t in tensors])


# --------------------------------------------------------------------------------------------------
# ROIAlign_v2  (operator_cxx/contrib/roi_align_v2{-inl.h,.cc,.cu})
# --------------------------------------------------------------------------------------------------
def roi_align_v2_forward(data, rois, pooled_size, spatial_scale):
    """_contrib_ROIAlign_v2 forward (roi_align_v2-inl.h:157-195).

    data (B,C,H,W), rois (B,R,4) -> output, maxidx_x, maxidx_y each (B,R,C,ph,pw)
    (shape inference roi_align_v2.cc:187-208).
    """
    _chk(data, "data", ndim=4)
    _chk(rois, "rois", ndim=3)
    if rois.shape[2] != 4:
        raise ValueError("bbox should be a 3D tensor of shape [batch, rois, 4]")
    if rois.shape[0] != data.shape[0]:
        raise ValueError("rois batch %d != data batch %d" % (rois.shape[0], data.shape[0]))
    ph, pw = _pair(pooled_size)
    B, C, H, W = data.shape
    R = rois.shape[1]
    shape = (B, R, C, ph, pw)
    out = torch.empty(shape, device=data.device, dtype=torch.float32)
    mx = torch.empty(shape, device=data.device, dtype=torch.float32)
    my = torch.empty(shape, device=data.device, dtype=torch.float32)
    lib().call("sd_roi_align_v2_fwd", _p(data), _p(rois), _p(out), _p(mx), _p(my), B, C, H, W, R,
               ph, pw, float(spatial_scale), _stream())
    return out, mx, my


def roi_align_v2_backward(out_grad, rois, maxidx_x, maxidx_y, data_shape, spatial_scale,
                          req_data="write", req_rois="write", d_data=None):
    """_backward_ROIAlign_v2 (roi_align_v2.cu:87-143): [dY, rois, maxidx_x, maxidx_y] -> [dX, d_rois]."""
    _chk(out_grad, "out_grad", ndim=5)
    _chk(rois, "rois", ndim=3)
    _chk(maxidx_x, "maxidx_x", ndim=5)
    _chk(maxidx_y, "maxidx_y", ndim=5)
    B, C, H, W = [int(v) for v in data_shape]
    Bo, R, Co, ph, pw = out_grad.shape
    if (Bo, Co) != (B, C) or rois.shape[:2] != (B, R):
        raise ValueError("shape mismatch between out_grad %s, rois %s and data %s"
                         % (tuple(out_grad.shape), tuple(rois.shape), (B, C, H, W)))
    rd = REQ[req_data] if isinstance(req_data, str) else int(req_data)
    rr = REQ[req_rois] if isinstance(req_rois, str) else int(req_rois)
    if d_data is None:
        if rd == REQ["add"]:
            raise ValueError("req_data='add' needs the d_data tensor to accumulate into")
        d_data = torch.empty((B, C, H, W), device=out_grad.device, dtype=torch.float32)
    _chk(d_data, "d_data", ndim=4)
    d_rois = torch.empty_like(rois) if rr != 0 else None
    lib().call("sd_roi_align_v2_bwd", _p(out_grad), _p(rois), _p(maxidx_x), _p(maxidx_y),
               _p(d_data), _p(d_rois), rd, rr, B, C, H, W, R, ph, pw, float(spatial_scale),
               _stream())
    return d_data, d_rois


# --------------------------------------------------------------------------------------------------
# FPN RoI extractor (models/FPN/builder.py:567-610) fused into one launch
# --------------------------------------------------------------------------------------------------
def fpn_roi_assign(rois, rcnn_stride, roi_canonical_scale=224, roi_canonical_level=4):
    """assign_layer_fpn CustomOp (models/FPN/assign_layer_fpn.py:17-41).

    rois (B,R,4) -> (list of len(rcnn_stride) zero-masked (B,R,4) tensors, level (B,R) int32)
    """
    _chk(rois, "rois", ndim=3)
    B, R, _ = rois.shape
    n = len(rcnn_stride)
    per = torch.empty((n, B, R, 4), device=rois.device, dtype=torch.float32)
    level = torch.empty((B, R), device=rois.device, dtype=torch.int32)
    lib().call("sd_fpn_roi_assign", _p(rois), B * R, _iarr(rcnn_stride), n,
               float(roi_canonical_scale), float(roi_canonical_level), _p(per), _p(level),
               _stream())
    return [per[i] for i in range(n)], level


def fpn_roi_align_forward(feats, rois, rcnn_stride, pooled_size, roi_canonical_scale=224,
                          roi_canonical_level=4):
    """FPNRoiAlign.get_roi_feature (models/FPN/builder.py:567-610) as one op.

    feats: list of (B,C,H_l,W_l); rois (B,R,4) -> out, maxidx_x, maxidx_y (B,R,C,ph,pw).
    """
    _chk(rois, "rois", ndim=3)
    if len(feats) != len(rcnn_stride):
        raise ValueError("one feature map per stride expected")
    B, C = feats[0].shape[:2]
    for i, f in enumerate(feats):
        _chk(f, "feats[%d]" % i, ndim=4)
        if tuple(f.shape[:2]) != (B, C):
            raise ValueError("all levels must share (B,C)")
    if rois.shape[0] != B:
        raise ValueError("rois batch mismatch")
    ph, pw = _pair(pooled_size)
    R = rois.shape[1]
    shape = (B, R, C, ph, pw)
    out = torch.empty(shape, device=rois.device, dtype=torch.float32)
    mx = torch.empty(shape, device=rois.device, dtype=torch.float32)
    my = torch.empty(shape, device=rois.device, dtype=torch.float32)
    lib().call("sd_fpn_roi_align_fwd", _parr(feats), _iarr([f.shape[2] for f in feats]),
               _iarr([f.shape[3] for f in feats]), _iarr(rcnn_stride), len(feats), _p(rois),
               _p(out), _p(mx), _p(my), B, C, R, ph, pw, float(roi_canonical_scale),
               float(roi_canonical_level), _stream())
    return out, mx, my


def fpn_roi_align_backward(out_grad, rois, maxidx_x, maxidx_y, feat_shapes, rcnn_stride,
                           roi_canonical_scale=224, roi_canonical_level=4, req_data="write",
                           d_feats=None):
    _chk(out_grad, "out_grad", ndim=5)
    _chk(rois, "rois", ndim=3)
    _chk(maxidx_x, "maxidx_x", ndim=5)
    _chk(maxidx_y, "maxidx_y", ndim=5)
    B, R, C, ph, pw = out_grad.shape
    rd = REQ[req_data] if isinstance(req_data, str) else int(req_data)
    if d_feats is None:
        if rd == REQ["add"]:
            raise ValueError("req_data='add' needs d_feats")
        d_feats = [torch.empty(tuple(s), device=out_grad.device, dtype=torch.float32)
                   for s in feat_shapes]
    for i, f in enumerate(d_feats):
        _chk(f, "d_feats[%d]" % i, ndim=4)
    lib().call("sd_fpn_roi_align_bwd", _p(out_grad), _p(rois), _p(maxidx_x), _p(maxidx_y),
               _parr(d_feats), _iarr([f.shape[2] for f in d_feats]),
               _iarr([f.shape[3] for f in d_feats]), _iarr(rcnn_stride), len(d_feats), rd, B, C, R,
               ph, pw, float(roi_canonical_scale), float(roi_canonical_level), _stream())
    return d_feats
