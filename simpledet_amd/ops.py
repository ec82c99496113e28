"""Torch-tensor harness over the C ABI (include/simpledet_ops.h).

PyTorch is plumbing here: it owns device memory and the HIP stream; every function passes raw
device pointers + the current stream to libsimpledet_ops_hip.so.  Nothing in this module computes
on the CPU and nothing falls back: non-CUDA tensors raise.

Function names/arguments mirror the reference operators (operator_cxx/, see each docstring).
"""
import ctypes

import torch

from ._lib import SD_ERR_UNSUPPORTED, SimpleDetOpsError, lib

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
    wsb = lib().cdll.sd_roi_align_v2_workspace_bytes(B, R)
    ws = torch.empty(wsb, device=data.device, dtype=torch.uint8)
    lib().call("sd_roi_align_v2_fwd_ws", _p(data), _p(rois), _p(out), _p(mx), _p(my), B, C, H, W, R,
               ph, pw, float(spatial_scale), _p(ws), ctypes.c_size_t(wsb), _stream())
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
    wsb = lib().cdll.sd_fpn_roi_align_workspace_bytes(B, R)
    ws = torch.empty(wsb, device=rois.device, dtype=torch.uint8)
    lib().call("sd_fpn_roi_align_fwd", _parr(feats), _iarr([f.shape[2] for f in feats]),
               _iarr([f.shape[3] for f in feats]), _iarr(rcnn_stride), len(feats), _p(rois),
               _p(out), _p(mx), _p(my), B, C, R, ph, pw, float(roi_canonical_scale),
               float(roi_canonical_level), _p(ws), ctypes.c_size_t(wsb), _stream())
    return out, mx, my


def argmax_stride(ph, pw):
    """bytes per (RoI, channel) row of the packed arg-max (sd_fpn_roi_align_argmax_stride)."""
    return int(lib().cdll.sd_fpn_roi_align_argmax_stride(int(ph), int(pw)))


def argmax_codes(argmax, pooled_size):
    """(B,R,C,S) packed arg-max -> (B,R,C,ph,pw) codes (drops the row padding)."""
    ph, pw = _pair(pooled_size)
    return argmax[..., :ph * pw].reshape(tuple(argmax.shape[:3]) + (ph, pw))


def fpn_roi_align_forward_packed(feats, rois, rcnn_stride, pooled_size, roi_canonical_scale=224,
                                 roi_canonical_level=4, plan=False):
    """The fused extractor with a one-byte arg-max: -> out (B,R,C,ph,pw) fp32, argmax (B,R,C,S)
    uint8 (S = ph*pw rounded up to a multiple of 4; code = row sample * 3 + column sample, 255 =
    nothing pooled; unpack with argmax_codes()), coords (B,R,9*(ph+pw)) 4-byte
    words: per RoI 3*(ph+pw) fp32 sample coordinates, then 3*(ph+pw) {neighbours, fraction} pairs.  (argmax, coords) are state between this op's forward and backward
    only; fpn_roi_align_backward_packed decodes them."""
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
    amax = torch.empty((B, R, C, argmax_stride(ph, pw)), device=rois.device, dtype=torch.uint8)
    coords = torch.empty((B, R, 9 * (ph + pw)), device=rois.device, dtype=torch.float32)
    wsb = lib().cdll.sd_fpn_roi_align_workspace_bytes(B, R)
    ws = torch.empty(wsb, device=rois.device, dtype=torch.uint8)
    if plan:
        # one rois-only pre-pass for the whole step: the backward's band lists / tap tables are built
        # here too and travel to fpn_roi_align_backward_packed as the third element of the state
        Hs, Ws = _iarr([f.shape[2] for f in feats]), _iarr([f.shape[3] for f in feats])
        pb = lib().cdll.sd_fpn_roi_align_plan_bytes(Hs, Ws, len(feats), B, R)
        plan_buf = torch.empty(pb, device=rois.device, dtype=torch.uint8)
        lib().call("sd_fpn_roi_align_fwd_packed_plan", _parr(feats), Hs, Ws, _iarr(rcnn_stride), len(feats),
                   _p(rois), _p(out), _p(amax), _p(coords), B, C, R, ph, pw, float(roi_canonical_scale),
                   float(roi_canonical_level), _p(ws), ctypes.c_size_t(wsb), _p(plan_buf),
                   ctypes.c_size_t(pb), _stream())
        return out, (amax, coords, plan_buf)
    lib().call("sd_fpn_roi_align_fwd_packed", _parr(feats), _iarr([f.shape[2] for f in feats]),
               _iarr([f.shape[3] for f in feats]), _iarr(rcnn_stride), len(feats), _p(rois),
               _p(out), _p(amax), _p(coords), B, C, R, ph, pw, float(roi_canonical_scale),
               float(roi_canonical_level), _p(ws), ctypes.c_size_t(wsb), _stream())
    return out, (amax, coords)


def fpn_roi_align_forward_packed_f16(feats, rois, rcnn_stride, pooled_size, roi_canonical_scale=224,
                                     roi_canonical_level=4):
    """fp16 I/O variant of fpn_roi_align_forward_packed: feats fp16 (B,C,H_l,W_l), rois fp32 (B,R,4)
    -> out fp16 (B,R,C,ph,pw), (argmax, coords).  Bit-equal to feats.float() -> the fp32 op ->
    out.half() (models/FPN/builder.py:581-586, 607-608 wraps the op in exactly those casts)."""
    _chk(rois, "rois", ndim=3)
    if len(feats) != len(rcnn_stride):
        raise ValueError("one feature map per stride expected")
    B, C = feats[0].shape[:2]
    for i, f in enumerate(feats):
        _chk(f, "feats[%d]" % i, dtype=torch.float16, ndim=4)
        if tuple(f.shape[:2]) != (B, C):
            raise ValueError("all levels must share (B,C)")
    if rois.shape[0] != B:
        raise ValueError("rois batch mismatch")
    ph, pw = _pair(pooled_size)
    R = rois.shape[1]
    out = torch.empty((B, R, C, ph, pw), device=rois.device, dtype=torch.float16)
    amax = torch.empty((B, R, C, argmax_stride(ph, pw)), device=rois.device, dtype=torch.uint8)
    coords = torch.empty((B, R, 9 * (ph + pw)), device=rois.device, dtype=torch.float32)
    wsb = lib().cdll.sd_fpn_roi_align_workspace_bytes(B, R)
    ws = torch.empty(wsb, device=rois.device, dtype=torch.uint8)
    try:
        lib().call("sd_fpn_roi_align_fwd_packed_f16", _parr(feats), _iarr([f.shape[2] for f in feats]),
                   _iarr([f.shape[3] for f in feats]), _iarr(rcnn_stride), len(feats), _p(rois),
                   _p(out), _p(amax), _p(coords), B, C, R, ph, pw, float(roi_canonical_scale),
                   float(roi_canonical_level), _p(ws), ctypes.c_size_t(wsb), _stream())
    except SimpleDetOpsError as e:
        if e.code != SD_ERR_UNSUPPORTED:
            raise
        # shapes the band-resident kernel does not take (too many units, W < 2, ...): the graph's own
        # cast -> fp32 op -> cast (models/FPN/builder.py:581-586, 607-608), same bits
        f32 = [cast_f16_to_f32(f) for f in feats]
        o32, (amax, coords) = fpn_roi_align_forward_packed(f32, rois, rcnn_stride, pooled_size,
                                                           roi_canonical_scale, roi_canonical_level)
        cast_f32_to_f16(o32, out)
    return out, (amax, coords)


def fpn_roi_align_backward_packed(out_grad, rois, argmax, feat_shapes, rcnn_stride,
                                  roi_canonical_scale=224, roi_canonical_level=4, req_data="write",
                                  d_feats=None):
    _chk(out_grad, "out_grad", ndim=5)
    _chk(rois, "rois", ndim=3)
    plan_buf = argmax[2] if len(argmax) > 2 else None   # forward(plan=True): lists / tap tables are built
    argmax, coords = argmax[0], argmax[1]
    _chk(argmax, "argmax", dtype=torch.uint8, ndim=4)
    _chk(coords, "coords", ndim=3)
    B, R, C, ph, pw = out_grad.shape
    if tuple(argmax.shape) != (B, R, C, argmax_stride(ph, pw)):
        raise ValueError("argmax must be (B,R,C,%d) uint8" % argmax_stride(ph, pw))
    rd = REQ[req_data] if isinstance(req_data, str) else int(req_data)
    if d_feats is None:
        if rd == REQ["add"]:
            raise ValueError("req_data='add' needs d_feats")
        d_feats = [torch.empty(tuple(s), device=out_grad.device, dtype=torch.float32)
                   for s in feat_shapes]
    for i, f in enumerate(d_feats):
        _chk(f, "d_feats[%d]" % i, ndim=4)
    hs, ws_ = _iarr([f.shape[2] for f in d_feats]), _iarr([f.shape[3] for f in d_feats])
    if plan_buf is not None:
        lib().call("sd_fpn_roi_align_bwd_packed_plan", _p(out_grad), _p(rois), _p(argmax), _p(coords),
                   _parr(d_feats), hs, ws_, _iarr(rcnn_stride), len(d_feats), rd, B, C, R, ph, pw,
                   float(roi_canonical_scale), float(roi_canonical_level), _p(plan_buf),
                   ctypes.c_size_t(plan_buf.numel()), _stream())
        return d_feats
    lib().cdll.sd_fpn_roi_align_bwd_workspace_bytes.restype = ctypes.c_size_t
    wsb = lib().cdll.sd_fpn_roi_align_bwd_workspace_bytes(hs, ws_, len(d_feats), B, R)
    work = torch.empty((max(int(wsb), 4) + 3) // 4, device=out_grad.device, dtype=torch.int32)
    lib().call("sd_fpn_roi_align_bwd_packed_ws", _p(out_grad), _p(rois), _p(argmax), _p(coords),
               _parr(d_feats), hs, ws_,
               _iarr(rcnn_stride), len(d_feats), rd, B, C, R, ph, pw, float(roi_canonical_scale),
               float(roi_canonical_level), _p(work), ctypes.c_size_t(work.numel() * 4), _stream())
    return d_feats


def cast_f16_to_f32(src, out=None):
    _chk(src, "src", dtype=torch.float16)
    out = torch.empty(src.shape, device=src.device, dtype=torch.float32) if out is None else out
    lib().call("sd_cast_f16_to_f32", _p(src), _p(out), ctypes.c_size_t(src.numel()), _stream())
    return out


def cast_f32_to_f16(src, out=None, req="write"):
    _chk(src, "src")
    out = torch.empty(src.shape, device=src.device, dtype=torch.float16) if out is None else out
    lib().call("sd_cast_f32_to_f16", _p(src), _p(out), ctypes.c_size_t(src.numel()),
               REQ[req] if isinstance(req, str) else int(req), _stream())
    return out


def fpn_roi_align_backward_packed_f16(out_grad, rois, argmax, feat_shapes, rcnn_stride,
                                      roi_canonical_scale=224, roi_canonical_level=4, req_data="write",
                                      d_feats=None, native=True):
    """Backward of fpn_roi_align_forward_packed_f16: out_grad fp16 -> gradients fp16, by the backward
    kernel's fp16-I/O instance (sd_fpn_roi_align_bwd_packed_f16: fp32 tap values, fixed-point sums,
    fp16 only at the two ends).  native=False, or a shape the wide kernel does not take: the two
    op-boundary casts of the reference's fp16 graphs around the fp32 kernel (same bits)."""
    _chk(out_grad, "out_grad", dtype=torch.float16, ndim=5)
    _chk(rois, "rois", ndim=3)
    rd = REQ[req_data] if isinstance(req_data, str) else int(req_data)
    if d_feats is None:
        if rd == REQ["add"]:
            raise ValueError("req_data='add' needs d_feats")
        d_feats = [torch.empty(tuple(s), device=out_grad.device, dtype=torch.float16) for s in feat_shapes]
    for d in d_feats:
        _chk(d, "d_feats", dtype=torch.float16, ndim=4)
    if native:
        am, coords = argmax[0], argmax[1]
        B, R, C, ph, pw = out_grad.shape
        hs, ws_ = _iarr([f.shape[2] for f in d_feats]), _iarr([f.shape[3] for f in d_feats])
        lib().cdll.sd_fpn_roi_align_bwd_workspace_bytes.restype = ctypes.c_size_t
        wsb = lib().cdll.sd_fpn_roi_align_bwd_workspace_bytes(hs, ws_, len(d_feats), B, R)
        work = torch.empty((max(int(wsb), 4) + 3) // 4, device=out_grad.device, dtype=torch.int32)
        try:
            lib().call("sd_fpn_roi_align_bwd_packed_f16", _p(out_grad), _p(rois), _p(am), _p(coords),
                       _parr(d_feats), hs, ws_, _iarr(rcnn_stride), len(d_feats), rd, B, C, R, ph, pw,
                       float(roi_canonical_scale), float(roi_canonical_level), _p(work),
                       ctypes.c_size_t(work.numel() * 4), _stream())
            return d_feats
        except SimpleDetOpsError as e:
            if e.code != SD_ERR_UNSUPPORTED:
                raise
    g32 = fpn_roi_align_backward_packed(cast_f16_to_f32(out_grad), rois, argmax, feat_shapes, rcnn_stride,
                                        roi_canonical_scale, roi_canonical_level)
    for g, d in zip(g32, d_feats):
        cast_f32_to_f16(g, d, rd)
    return d_feats


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


# --------------------------------------------------------------------------------------------------
# ROIPooling_v1  (operator_cxx/roi_pooling_v1{-inl.h,.cc,.cu})
# --------------------------------------------------------------------------------------------------
def roi_pool_v1_forward(data, rois, pooled_size, spatial_scale):
    """ROIPooling_v1 forward: data (B,C,H,W), rois (K,5) -> output, maxidx (K,C,ph,pw)
    (shape inference roi_pooling_v1-inl.h:172-199)."""
    _chk(data, "data", ndim=4)
    _chk(rois, "rois", ndim=2)
    if rois.shape[1] != 5:
        raise ValueError("bbox should be a 2D tensor of shape [batch, 5]")
    ph, pw = _pair(pooled_size)
    B, C, H, W = data.shape
    K = rois.shape[0]
    out = torch.empty((K, C, ph, pw), device=data.device, dtype=torch.float32)
    idx = torch.empty_like(out)
    lib().call("sd_roi_pool_v1_fwd", _p(data), _p(rois), _p(out), _p(idx), B, C, H, W, K, ph, pw,
               float(spatial_scale), _stream())
    return out, idx


def roi_pool_v1_backward(out_grad, rois, maxidx, data_shape, spatial_scale, req_data="write",
                         req_rois="write", d_data=None):
    """_backward_ROIPooling_v1: [dY, rois, maxidx] -> [dX, d_rois] (roi_pooling_v1-inl.h:96-133)."""
    _chk(out_grad, "out_grad", ndim=4)
    _chk(rois, "rois", ndim=2)
    _chk(maxidx, "maxidx", ndim=4)
    B, C, H, W = [int(v) for v in data_shape]
    K, Co, ph, pw = out_grad.shape
    if Co != C or rois.shape[0] != K:
        raise ValueError("shape mismatch")
    rd = REQ[req_data] if isinstance(req_data, str) else int(req_data)
    rr = REQ[req_rois] if isinstance(req_rois, str) else int(req_rois)
    if d_data is None:
        if rd == REQ["add"]:
            raise ValueError("req_data='add' needs the d_data tensor to accumulate into")
        d_data = torch.empty((B, C, H, W), device=out_grad.device, dtype=torch.float32)
    d_rois = torch.empty_like(rois) if rr != 0 else None
    lib().call("sd_roi_pool_v1_bwd", _p(out_grad), _p(rois), _p(maxidx), _p(d_data), _p(d_rois), rd,
               rr, B, C, H, W, K, ph, pw, float(spatial_scale), _stream())
    return d_data, d_rois


# --------------------------------------------------------------------------------------------------
# _contrib_GenAnchor  (operator_cxx/contrib/generate_anchor{-inl.h,.cc,.cu})
# --------------------------------------------------------------------------------------------------
def _darr(vals):
    return (ctypes.c_double * len(vals))(*[float(v) for v in vals])


def gen_anchor(height, width, feature_stride, scales, ratios, device=None):
    """GenAnchor: (H*W*A, 4) fp32 anchors, row (h*W + w)*A + a, A ratio-major
    (generate_anchor-inl.h:176-180; shape generate_anchor-inl.h:92-106)."""
    scales, ratios = list(scales), list(ratios)
    device = device or torch.device("cuda", torch.cuda.current_device())
    A = len(scales) * len(ratios)
    out = torch.empty((int(height) * int(width) * A, 4), device=device, dtype=torch.float32)
    lib().call("sd_gen_anchor", _p(out), int(height), int(width), int(feature_stride),
               _darr(scales), len(scales), _darr(ratios), len(ratios), _stream())
    return out


def gen_anchor_levels(shapes, strides, scales, ratios, device=None):
    """GenAnchor for every pyramid level in one launch: shapes [(H, W), ...], strides [...] ->
    list of (H*W*A, 4) tensors (each equal to gen_anchor of that level)."""
    scales, ratios = list(scales), list(ratios)
    device = device or torch.device("cuda", torch.cuda.current_device())
    A = len(scales) * len(ratios)
    outs = [torch.empty((int(h) * int(w) * A, 4), device=device, dtype=torch.float32) for h, w in shapes]
    n = len(outs)
    ptrs = (ctypes.c_void_p * n)(*[o.data_ptr() for o in outs])
    hs = (ctypes.c_int * n)(*[int(h) for h, _ in shapes])
    ws = (ctypes.c_int * n)(*[int(w) for _, w in shapes])
    st = (ctypes.c_int * n)(*[int(v) for v in strides])
    lib().call("sd_gen_anchor_levels", ptrs, hs, ws, st, n, _darr(scales), len(scales), _darr(ratios),
               len(ratios), _stream())
    return outs


# --------------------------------------------------------------------------------------------------
# ProposalTarget  (operator_cxx/proposal_target{-inl.h,.cc})
# --------------------------------------------------------------------------------------------------
class ProposalTargetParam(ctypes.Structure):
    """sd_proposal_target_param == ProposalTargetParam (proposal_target-inl.h:81-114)."""
    _fields_ = [("num_classes", ctypes.c_int), ("batch_images", ctypes.c_int),
                ("image_rois", ctypes.c_int), ("fg_fraction", ctypes.c_float),
                ("fg_thresh", ctypes.c_float), ("bg_thresh_hi", ctypes.c_float),
                ("bg_thresh_lo", ctypes.c_float), ("proposal_without_gt", ctypes.c_int),
                ("class_agnostic", ctypes.c_int), ("bbox_mean", ctypes.c_float * 4),
                ("bbox_std", ctypes.c_float * 4), ("bbox_weight", ctypes.c_float * 4)]


def glibc_rand_state(seed=1, device=None):
    """Device copy of libc's rand() state after srand(seed) (33 int32).  seed=1 is the state of a
    process that never called srand -- what the reference op sees (no srand anywhere in it)."""
    host = (ctypes.c_int32 * 33)()
    lib().call("sd_glibc_srand_host", ctypes.c_uint32(seed), host)
    device = device or torch.device("cuda", torch.cuda.current_device())
    return torch.tensor(list(host), dtype=torch.int32, device=device)


_default_rng = {}


def default_rng_state(device=None, reset_seed=None):
    """The process-wide (per device) glibc rand() state ProposalTarget advances when the caller
    passes none -- libc's global state in the reference.  reset_seed re-seeds it (srand)."""
    device = torch.device(device) if device is not None else torch.device("cuda", torch.cuda.current_device())
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if reset_seed is not None or device not in _default_rng:
        _default_rng[device] = glibc_rand_state(1 if reset_seed is None else reset_seed, device)
    return _default_rng[device]


def proposal_target(rois, gt_boxes, num_classes, batch_images, image_rois, fg_fraction=0.25,
                    fg_thresh=0.5, bg_thresh_hi=0.5, bg_thresh_lo=0.0, proposal_without_gt=False,
                    class_agnostic=False, bbox_mean=(0., 0., 0., 0.), bbox_std=(.1, .1, .2, .2),
                    bbox_weight=(1., 1., 1., 1.), rng_state=None, return_index=False,
                    valid_ranges=None, filter_scales=False):
    """ProposalTarget: rois (B,N,4), gt_boxes (B,M,5) -> roi_output (B,S,4), label (B,S),
    bbox_target (B,S,4K), bbox_weight (B,S,4K), match_gt_iou (B,S)  (proposal_target-inl.h:297-330).
    rng_state: int32[33] device tensor from glibc_rand_state(); advanced in place.
    valid_ranges (B,2) selects ProposalTarget_v2 (proposal_target_v2-inl.h), with filter_scales."""
    _chk(rois, "rois", ndim=3)
    _chk(gt_boxes, "gt_boxes", ndim=3)
    B, N, _ = rois.shape
    if rois.shape[2] != 4 or gt_boxes.shape[2] != 5 or gt_boxes.shape[0] != B:
        raise ValueError("rois must be (B,N,4) and gt_boxes (B,M,5)")
    if B != int(batch_images):
        raise ValueError("batch_images=%d but rois has batch %d" % (batch_images, B))
    M = gt_boxes.shape[1]
    p = ProposalTargetParam()
    p.num_classes, p.batch_images, p.image_rois = int(num_classes), int(batch_images), int(image_rois)
    p.fg_fraction, p.fg_thresh = float(fg_fraction), float(fg_thresh)
    p.bg_thresh_hi, p.bg_thresh_lo = float(bg_thresh_hi), float(bg_thresh_lo)
    p.proposal_without_gt, p.class_agnostic = int(bool(proposal_without_gt)), int(bool(class_agnostic))
    for i in range(4):
        p.bbox_mean[i], p.bbox_std[i], p.bbox_weight[i] = bbox_mean[i], bbox_std[i], bbox_weight[i]
    if rng_state is None:
        # the reference never calls srand: libc's global state starts at seed 1 and ADVANCES from
        # call to call, so every step draws a fresh stretch of the rand() stream.  Keep one such
        # state per device for callers that do not manage their own (mx.sym.ProposalTarget has no
        # rng argument).
        rng_state = default_rng_state(rois.device)
    _chk(rng_state, "rng_state", dtype=torch.int32, ndim=1)
    if rng_state.numel() != 33:
        raise ValueError("rng_state must hold 33 int32 words (glibc_rand_state), got %d"
                         % rng_state.numel())
    S, K4 = int(image_rois), 4 * int(num_classes)
    dev = rois.device
    ro = torch.empty((B, S, 4), device=dev, dtype=torch.float32)
    lb = torch.empty((B, S), device=dev, dtype=torch.float32)
    bt = torch.empty((B, S, K4), device=dev, dtype=torch.float32)
    bw = torch.empty((B, S, K4), device=dev, dtype=torch.float32)
    iou = torch.empty((B, S), device=dev, dtype=torch.float32)
    kept = torch.empty((B, S), device=dev, dtype=torch.int32) if return_index else None
    wsb = lib().cdll.sd_proposal_target_workspace_bytes(B, N, M)
    ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
    if valid_ranges is not None:
        _chk(valid_ranges, "valid_ranges", ndim=2)
        if tuple(valid_ranges.shape) != (B, 2):
            raise ValueError("valid_ranges must be (B,2)")
        lib().call("sd_proposal_target_v2", _p(rois), _p(gt_boxes), _p(valid_ranges),
                   int(bool(filter_scales)), N, M, ctypes.byref(p), _p(rng_state), _p(ro), _p(lb),
                   _p(bt), _p(bw), _p(iou), _p(kept), _p(ws), ctypes.c_size_t(wsb), _stream())
    else:
        lib().call("sd_proposal_target", _p(rois), _p(gt_boxes), N, M, ctypes.byref(p), _p(rng_state),
                   _p(ro), _p(lb), _p(bt), _p(bw), _p(iou), _p(kept), _p(ws), ctypes.c_size_t(wsb),
                   _stream())
    res = (ro, lb, bt, bw, iou)
    return res + (kept,) if return_index else res


def proposal_mask_target(rois, gt_boxes, gt_polys, num_classes, batch_images, image_rois, mask_size=28,
                         fg_fraction=0.25, fg_thresh=0.5, bg_thresh_hi=0.5, bg_thresh_lo=0.0,
                         proposal_without_gt=False, class_agnostic=False, bbox_mean=(0., 0., 0., 0.),
                         bbox_std=(.1, .1, .2, .2), bbox_weight=(1., 1., 1., 1.), rng_state=None,
                         valid_ranges=None, filter_scales=False, return_index=False, output_ratio=False,
                         max_raster_pixels=1408 * 1408):
    """ProposalMaskTarget (proposal_mask_target-inl.h): ProposalTarget_v2's five outputs plus
    mask_target (B, int(image_rois*fg_fraction), mask_size, mask_size): the 0/1 mask of each sampled
    foreground RoI's gt polygon in the RoI's frame, -1 rows past the sampled foreground.
    gt_polys (B,M,L): [category, n_seg, len_1..len_n, x,y,...] padded with -1.
    output_ratio=True (mask scoring R-CNN, proposal_mask_target.cc:20-152) appends mask_ratio (B, FG)
    after mask_target; max_raster_pixels bounds the image-resolution rasters it counts (a row whose
    RoI or RoI-and-polygon bounding box has more pixels gets NaN)."""
    _chk(rois, "rois", ndim=3)
    _chk(gt_boxes, "gt_boxes", ndim=3)
    _chk(gt_polys, "gt_polys", ndim=3)
    B, N, _ = rois.shape
    M, L = gt_boxes.shape[1], gt_polys.shape[2]
    if gt_polys.shape[:2] != gt_boxes.shape[:2] or B != int(batch_images):
        raise ValueError("gt_polys must be (B,M,L) like gt_boxes (B,M,5), B = batch_images")
    p = ProposalTargetParam()
    p.num_classes, p.batch_images, p.image_rois = int(num_classes), int(batch_images), int(image_rois)
    p.fg_fraction, p.fg_thresh = float(fg_fraction), float(fg_thresh)
    p.bg_thresh_hi, p.bg_thresh_lo = float(bg_thresh_hi), float(bg_thresh_lo)
    p.proposal_without_gt, p.class_agnostic = int(bool(proposal_without_gt)), int(bool(class_agnostic))
    for i in range(4):
        p.bbox_mean[i], p.bbox_std[i], p.bbox_weight[i] = bbox_mean[i], bbox_std[i], bbox_weight[i]
    if rng_state is None:
        rng_state = default_rng_state(rois.device)
    _chk(rng_state, "rng_state", dtype=torch.int32, ndim=1)
    if rng_state.numel() != 33:
        raise ValueError("rng_state must hold 33 int32 words")
    if valid_ranges is not None:
        _chk(valid_ranges, "valid_ranges", ndim=2)
    S, K4 = int(image_rois), 4 * int(num_classes)
    FG = int(torch.tensor(S, dtype=torch.float32) * torch.tensor(fg_fraction, dtype=torch.float32))
    dev = rois.device
    ro = torch.empty((B, S, 4), device=dev, dtype=torch.float32)
    lb = torch.empty((B, S), device=dev, dtype=torch.float32)
    bt = torch.empty((B, S, K4), device=dev, dtype=torch.float32)
    bw = torch.empty((B, S, K4), device=dev, dtype=torch.float32)
    iou = torch.empty((B, S), device=dev, dtype=torch.float32)
    mask = torch.empty((B, FG, int(mask_size), int(mask_size)), device=dev, dtype=torch.float32)
    kept = torch.empty((B, S), device=dev, dtype=torch.int32) if return_index else None
    if output_ratio:
        ratio = torch.empty((B, FG), device=dev, dtype=torch.float32)
        wsb = lib().cdll.sd_proposal_mask_target_ratio_workspace_bytes(
            B, N, M, S, ctypes.c_float(float(fg_fraction)), int(max_raster_pixels))
        ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
        lib().call("sd_proposal_mask_target_ratio", _p(rois), _p(gt_boxes), _p(gt_polys), _p(valid_ranges),
                   int(bool(filter_scales)), N, M, L, int(mask_size), ctypes.byref(p), _p(rng_state), _p(ro),
                   _p(lb), _p(bt), _p(bw), _p(iou), _p(mask), _p(ratio), int(max_raster_pixels), _p(kept),
                   _p(ws), ctypes.c_size_t(wsb), _stream())
        res = (ro, lb, bt, bw, iou, mask, ratio)
        return res + (kept,) if return_index else res
    wsb = lib().cdll.sd_proposal_target_workspace_bytes(B, N, M)
    ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
    lib().call("sd_proposal_mask_target", _p(rois), _p(gt_boxes), _p(gt_polys), _p(valid_ranges),
               int(bool(filter_scales)), N, M, L, int(mask_size), ctypes.byref(p), _p(rng_state), _p(ro),
               _p(lb), _p(bt), _p(bw), _p(iou), _p(mask), _p(kept), _p(ws), ctypes.c_size_t(wsb), _stream())
    res = (ro, lb, bt, bw, iou, mask)
    return res + (kept,) if return_index else res


# --------------------------------------------------------------------------------------------------
# RPN anchor-target assignment (core/detection_input.py:345-565, models/FPN/input.py:9-146)
# --------------------------------------------------------------------------------------------------
class RpnTargetParam(ctypes.Structure):
    """sd_rpn_target_param"""
    _fields_ = [("nlvl", ctypes.c_int), ("stride", ctypes.c_int * 8), ("short_side", ctypes.c_int * 8),
                ("long_side", ctypes.c_int * 8), ("n_scales", ctypes.c_int), ("n_aspects", ctypes.c_int),
                ("scales", ctypes.c_double * 16), ("aspects", ctypes.c_double * 16),
                ("allowed_border", ctypes.c_int), ("pos_thr", ctypes.c_float), ("neg_thr", ctypes.c_float),
                ("min_pos_thr", ctypes.c_float), ("image_anchor", ctypes.c_int),
                ("pos_fraction", ctypes.c_double)]


def _seq(v):
    return list(v) if isinstance(v, (tuple, list)) else [v]


def rpn_target_param(stride, short, long, scales, aspects, allowed_border=0, pos_thr=0.7, neg_thr=0.3,
                     min_pos_thr=0.0, image_anchor=256, pos_fraction=0.5):
    """the AnchorTarget2DParam of a config (config/faster_r50v1_fpn_1x.py:212-232) as the C struct"""
    p = RpnTargetParam()
    st, sh, lg, sc, asp = _seq(stride), _seq(short), _seq(long), _seq(scales), _seq(aspects)
    if not (len(st) == len(sh) == len(lg)) or len(st) > 8 or len(sc) * len(asp) > 16:
        raise ValueError("bad pyramid / anchor description")
    p.nlvl, p.n_scales, p.n_aspects = len(st), len(sc), len(asp)
    for i in range(len(st)):
        p.stride[i], p.short_side[i], p.long_side[i] = int(st[i]), int(sh[i]), int(lg[i])
    for i, v in enumerate(sc):
        p.scales[i] = float(v)
    for i, v in enumerate(asp):
        p.aspects[i] = float(v)
    p.allowed_border = int(allowed_border)
    p.pos_thr, p.neg_thr, p.min_pos_thr = float(pos_thr), float(neg_thr), float(min_pos_thr)
    p.image_anchor, p.pos_fraction = int(image_anchor), float(pos_fraction)
    return p


def mt19937_state(seed=None, numpy_state=None, device=None):
    """Device copy (int32[625]) of a numpy RandomState: np.random.seed(seed), or an explicit
    RandomState.get_state() tuple."""
    device = device or torch.device("cuda", torch.cuda.current_device())
    host = (ctypes.c_int32 * 625)()
    if numpy_state is not None:
        key, pos = numpy_state[1], int(numpy_state[2])
        vals = [int(k) - (1 << 32) if int(k) >= (1 << 31) else int(k) for k in key] + [pos]
        return torch.tensor(vals, dtype=torch.int32, device=device)
    lib().call("sd_mt19937_seed_host", ctypes.c_uint32(int(seed)), host)
    return torch.tensor(list(host), dtype=torch.int32, device=device)


def rpn_anchor_target(im_info, gt_bbox, param, mt_state, layout=1):
    """RPN labels / box targets / weights for a batch of images, the reference loader's
    (Pyramid)AnchorTarget2D on the device.  im_info (B,3), gt_bbox (B,M,4|5) padded with -1 rows.
    layout 1: cls_label (B, A*sumHW), reg_target / reg_weight (B, 4A, sumHW); layout 0: (B,N), (B,N,4).
    mt_state: mt19937_state(...), advanced in place like np.random."""
    _chk(im_info, "im_info", ndim=2)
    _chk(gt_bbox, "gt_bbox", ndim=3)
    _chk(mt_state, "mt_state", dtype=torch.int32, ndim=1)
    if mt_state.numel() != 625:
        raise ValueError("mt_state must hold 625 int32 words")
    B, M, G = gt_bbox.shape
    N = int(lib().cdll.sd_rpn_target_num_anchors(ctypes.byref(param)))
    if N < 0:
        raise ValueError("bad rpn target parameters")
    A = param.n_scales * param.n_aspects
    dev = im_info.device
    if layout == 1:
        cls = torch.empty((B, N), device=dev, dtype=torch.float32)
        tgt = torch.empty((B, 4 * A, N // A), device=dev, dtype=torch.float32)
        wgt = torch.empty_like(tgt)
    else:
        cls = torch.empty((B, N), device=dev, dtype=torch.float32)
        tgt = torch.empty((B, N, 4), device=dev, dtype=torch.float32)
        wgt = torch.empty_like(tgt)
    lib().cdll.sd_rpn_target_workspace_bytes.restype = ctypes.c_size_t
    wsb = lib().cdll.sd_rpn_target_workspace_bytes(ctypes.byref(param), B, M)
    ws = torch.empty(wsb, device=dev, dtype=torch.uint8)
    lib().call("sd_rpn_anchor_target", _p(im_info), _p(gt_bbox), B, M, G, ctypes.byref(param), _p(mt_state),
               _p(cls), _p(tgt), _p(wgt), int(layout), _p(ws), ctypes.c_size_t(wsb), _stream())
    return cls, tgt, wgt


# --------------------------------------------------------------------------------------------------
# _contrib_NMS  (operator_cxx/contrib/nms{-inl.h,.cu}) and the Cython soft-NMS family
# --------------------------------------------------------------------------------------------------
def nms(dets, rpn_pre_nms_top_n=6000, rpn_post_nms_top_n=300, threshold=0.7, already_sorted=False,
        threshold_ge=False, return_index=False):
    """_contrib_NMS (GPU path nms.cu:249-365): dets (B,N,5) -> out (B,post,4), score (B,post,1)
    (shape nms-inl.h:96-107; post = min(rpn_post_nms_top_n, pre))."""
    _chk(dets, "dets", ndim=3)
    if dets.shape[2] != 5:
        raise ValueError("bbox should be (batch, rois, 5)")
    B, N, _ = dets.shape
    pre = int(rpn_pre_nms_top_n) if rpn_pre_nms_top_n > 0 else N
    pre = min(pre, N)
    post = min(int(rpn_post_nms_top_n), pre)
    out = torch.empty((B, post, 4), device=dets.device, dtype=torch.float32)
    score = torch.empty((B, post, 1), device=dets.device, dtype=torch.float32)
    keep = torch.empty((B, post), device=dets.device, dtype=torch.int32) if return_index else None
    wsb = lib().cdll.sd_nms_workspace_bytes(B, N, int(rpn_pre_nms_top_n))
    ws = torch.empty(wsb, device=dets.device, dtype=torch.uint8)
    lib().call("sd_nms", _p(dets), B, N, int(rpn_pre_nms_top_n), int(rpn_post_nms_top_n),
               float(threshold), int(bool(threshold_ge)), int(bool(already_sorted)), _p(out),
               _p(score), _p(keep), _p(ws), ctypes.c_size_t(wsb), _stream())
    return (out, score, keep) if return_index else (out, score)


SOFT_NMS_METHODS = {"hard": 0, "linear": 1, "gaussian": 2}


def soft_nms_batched(dets, counts=None, sigma=0.5, Nt=0.3, threshold=0.001, method=0):
    """soft_nms (cpu_nms.pyx:98-203) on P problems at once: dets (P,Nmax,5), counts (P) int32 ->
    out_dets (P,Nmax,5), out_inds (P,Nmax) int32, out_counts (P) int32 (rows past the count are
    unspecified)."""
    _chk(dets, "dets", ndim=3)
    if dets.shape[2] != 5:
        raise ValueError("dets should be (problems, boxes, 5)")
    P, Nmax, _ = dets.shape
    if counts is not None:
        _chk(counts, "counts", dtype=torch.int32, ndim=1)
    if isinstance(method, str):
        if method not in SOFT_NMS_METHODS:
            raise ValueError("Unknown soft_nms method: {}".format(method))
        method = SOFT_NMS_METHODS[method]
    od = torch.empty_like(dets)
    oi = torch.empty((P, Nmax), device=dets.device, dtype=torch.int32)
    oc = torch.empty((P,), device=dets.device, dtype=torch.int32)
    lib().call("sd_soft_nms_batched", _p(dets), _p(counts), P, Nmax, float(sigma), float(Nt),
               float(threshold), int(method), _p(od), _p(oi), _p(oc), _stream())
    return od, oi, oc


def bbox_overlaps(boxes, query_boxes):
    """bbox_overlaps_cython (bbox.pyx:31-72): boxes (n,4), query_boxes (k,4) -> overlaps (n,k)."""
    _chk(boxes, "boxes", ndim=2)
    _chk(query_boxes, "query_boxes", ndim=2)
    n, k = boxes.shape[0], query_boxes.shape[0]
    ov = torch.empty((n, k), device=boxes.device, dtype=torch.float32)
    lib().call("sd_bbox_overlaps", _p(boxes), n, _p(query_boxes), k, _p(ov), _stream())
    return ov


# --------------------------------------------------------------------------------------------------
# DeformableConvolution v1 (mx.sym.contrib.DeformableConvolution, models/dcn/builder.py:14-17)
# --------------------------------------------------------------------------------------------------
def _dcn_out_hw(H, W, kh, kw, pad, stride, dil):
    return ((H + 2 * pad - (dil * (kh - 1) + 1)) // stride + 1,
            (W + 2 * pad - (dil * (kw - 1) + 1)) // stride + 1)


def deform_im2col(x, offset, kernel=(3, 3), pad=1, stride=1, dilate=1, num_deformable_group=1):
    """deformable_im2col: x (N,C,H,W), offset (N,dg*2*kh*kw,Ho,Wo) -> col (N, C*kh*kw, Ho*Wo)."""
    _chk(x, "x", ndim=4)
    _chk(offset, "offset", ndim=4)
    N, C, H, W = x.shape
    kh, kw = kernel
    Ho, Wo = _dcn_out_hw(H, W, kh, kw, pad, stride, dilate)
    col = torch.empty((N, C * kh * kw, Ho * Wo), device=x.device, dtype=torch.float32)
    lib().call("sd_deform_im2col", _p(x), _p(offset), _p(col), N, C, H, W, kh, kw, pad, pad, stride,
               stride, dilate, dilate, int(num_deformable_group), _stream())
    return col


def deform_col2im(col, offset, x_shape, kernel=(3, 3), pad=1, stride=1, dilate=1,
                  num_deformable_group=1, workspace=True):
    _chk(col, "col", ndim=3)
    _chk(offset, "offset", ndim=4)
    N, C, H, W = [int(v) for v in x_shape]
    kh, kw = kernel
    dx = torch.empty((N, C, H, W), device=col.device, dtype=torch.float32)
    if not workspace:   # fp32 compare-and-swap adds (what a caller without a workspace gets)
        lib().call("sd_deform_col2im", _p(col), _p(offset), _p(dx), REQ["write"], N, C, H, W, kh, kw, pad,
                   pad, stride, stride, dilate, dilate, int(num_deformable_group), _stream())
        return dx
    lib().cdll.sd_deform_col2im_workspace_bytes.restype = ctypes.c_size_t
    n = int(lib().cdll.sd_deform_col2im_workspace_bytes(N, int(num_deformable_group)))
    ws = torch.empty(n, device=col.device, dtype=torch.uint8)
    lib().call("sd_deform_col2im_ws", _p(col), _p(offset), _p(dx), REQ["write"], N, C, H, W, kh, kw, pad,
               pad, stride, stride, dilate, dilate, int(num_deformable_group), _p(ws), ctypes.c_size_t(n),
               _stream())
    return dx


def deform_col2im_coord(col, x, offset, kernel=(3, 3), pad=1, stride=1, dilate=1,
                        num_deformable_group=1):
    _chk(col, "col", ndim=3)
    _chk(x, "x", ndim=4)
    _chk(offset, "offset", ndim=4)
    N, C, H, W = x.shape
    kh, kw = kernel
    doff = torch.empty_like(offset)
    lib().call("sd_deform_col2im_coord", _p(col), _p(x), _p(offset), _p(doff), REQ["write"], N, C, H,
               W, kh, kw, pad, pad, stride, stride, dilate, dilate, int(num_deformable_group),
               _stream())
    return doff


def gemm_f32(a, b, trans_a=False, trans_b=False, out=None, accumulate=0):
    """Batched fp32-in / fp32-out matrix-core GEMM: a (Bt,M,K) or its transpose, b (Bt,K,N) or its
    transpose.  Default arithmetic: scaled fp16 hi/lo split on the f16 matrix cores after a max|.|
    pre-pass over both operands (accuracy of the fp32 MFMA path); tuning key `deform_gemm_split` = 1:
    bf16 split (4.5e-6 x max|C|), 0: fp32 MFMA; see include/simpledet_ops.h."""
    _chk(a, "a", ndim=3)
    _chk(b, "b", ndim=3)
    Bt = a.shape[0]
    M, K = (a.shape[2], a.shape[1]) if trans_a else (a.shape[1], a.shape[2])
    K2, N = (b.shape[2], b.shape[1]) if trans_b else (b.shape[1], b.shape[2])
    if K != K2 or b.shape[0] != Bt:
        raise ValueError("GEMM shape mismatch")
    if out is None:
        out = torch.empty((Bt, M, N), device=a.device, dtype=torch.float32)
    ws = torch.empty(64, device=a.device, dtype=torch.uint8)
    lib().call("sd_gemm_f32_ws", int(trans_a), int(trans_b), M, N, K, _p(a), a.shape[2],
               a.shape[1] * a.shape[2], _p(b), b.shape[2], b.shape[1] * b.shape[2], _p(out), N,
               M * N, Bt, int(accumulate), _p(ws), ctypes.c_size_t(64), _stream())
    return out


def _dcn_ws(x, kh, kw, pad, stride, dilate):
    N, C, H, W = x.shape
    n = lib().cdll.sd_deform_conv_workspace_bytes(N, C, H, W, kh, kw, pad, stride, dilate)
    return torch.empty(n, device=x.device, dtype=torch.uint8), n


def deform_conv_forward(x, offset, weight, pad=1, stride=1, dilate=1, num_deformable_group=1,
                        keep_col=False, bias=None, num_group=1):
    """DeformableConvolution forward: y (N,F,Ho,Wo).  weight (F, C / num_group, kh, kw); bias (F) or None
    (= no_bias).  keep_col=True returns (y, workspace): the workspace holds the col matrix; handed to
    deform_conv_backward(fwd_ws=...) it saves the backward its own im2col."""
    _chk(x, "data", ndim=4)
    _chk(offset, "offset", ndim=4)
    _chk(weight, "weight", ndim=4)
    N, C, H, W = x.shape
    F, Cw, kh, kw = weight.shape
    num_group = int(num_group)
    if num_group < 1 or C % num_group or F % num_group:
        raise ValueError("num_group %d must divide data channels %d and num_filter %d" % (num_group, C, F))
    if Cw * num_group != C:
        raise ValueError("weight channels %d != data channels %d / num_group %d" % (Cw, C, num_group))
    if bias is not None:
        _chk(bias, "bias", ndim=1)
        if bias.shape[0] != F:
            raise ValueError("bias has %d entries, num_filter is %d" % (bias.shape[0], F))
    Ho, Wo = _dcn_out_hw(H, W, kh, kw, pad, stride, dilate)
    if tuple(offset.shape) != (N, num_deformable_group * 2 * kh * kw, Ho, Wo):
        raise ValueError("offset shape %s != %s" % (tuple(offset.shape),
                                                     (N, num_deformable_group * 2 * kh * kw, Ho, Wo)))
    y = torch.empty((N, F, Ho, Wo), device=x.device, dtype=torch.float32)
    # keep_col = False: no col matrix where the shape allows -- sampling fused into the GEMM (3x3, num_group 1,
    # C / groups % 16 == 0, H*W % 4 == 0); other shapes run im2col + GEMM behind the same entry point, which
    # its workspace size accounts for
    n = int(lib().cdll.sd_deform_convolution_fwd_workspace_bytes(N, C, H, W, F, kh, kw, pad, stride, dilate,
                                                                 int(num_deformable_group), num_group,
                                                                 int(bool(keep_col))))
    ws = torch.empty(n, device=x.device, dtype=torch.uint8)
    lib().call("sd_deform_convolution_fwd", _p(x), _p(offset), _p(weight), _p(bias) if bias is not None else None,
               _p(y), N, C, H, W, F, kh, kw, pad, stride, dilate, int(num_deformable_group), num_group,
               int(bool(keep_col)), _p(ws), ctypes.c_size_t(n), _stream())
    return (y, ws) if keep_col else y


def deform_conv_backward(out_grad, x, offset, weight, pad=1, stride=1, dilate=1,
                         num_deformable_group=1, req=("write", "write", "write"), grads=None,
                         fwd_ws=None, num_group=1, bias=False):
    """-> (d_data, d_offset, d_weight[, d_bias]).  fwd_ws: the workspace deform_conv_forward(keep_col=True)
    returned for the same (x, offset): its col matrix is reused.  bias=True (the op had a bias): a fourth
    req / gradient, d_bias (F) = sum over images and pixels of out_grad."""
    _chk(out_grad, "out_grad", ndim=4)
    N, C, H, W = x.shape
    F, _, kh, kw = weight.shape
    r = [REQ[v] if isinstance(v, str) else int(v) for v in req]
    if bias and len(r) == 3:
        r.append(REQ["write"])
    if grads is None:
        grads = (torch.empty_like(x), torch.empty_like(offset), torch.empty_like(weight))
        if bias:
            grads = grads + (torch.empty(F, device=x.device, dtype=torch.float32),)
    elif len(grads) != (4 if bias else 3):
        raise ValueError("grads must hold %d tensors (d_data, d_offset, d_weight%s), got %d"
                         % (4 if bias else 3, ", d_bias" if bias else "", len(grads)))
    ws, n = _dcn_ws(x, kh, kw, pad, stride, dilate)
    col = None
    if fwd_ws is not None:
        col = ctypes.c_void_p((fwd_ws.data_ptr() + 255) & ~255)  # sd_deform_conv_col_of_workspace
    lib().call("sd_deform_convolution_bwd", _p(out_grad), _p(x), _p(offset), _p(weight), col, _p(grads[0]),
               _p(grads[1]), _p(grads[2]), _p(grads[3]) if bias else None, r[0], r[1], r[2],
               r[3] if bias else REQ["null"], N, C, H, W, F, kh, kw, pad, stride, dilate,
               int(num_deformable_group), int(num_group), _p(ws), ctypes.c_size_t(n), _stream())
    return grads


# --------------------------------------------------------------------------------------------------
# _contrib_Proposal_v3 + get_top_proposal  (operator_cxx/contrib/proposal_v3{-inl.h,.cu},
# models/FPN/get_top_proposal.py) -- SURVEY 8(f) rank 1
# --------------------------------------------------------------------------------------------------
def _farr(vals):
    return (ctypes.c_float * len(vals))(*[float(v) for v in vals])


def proposal_v3(cls_prob, bbox_pred, im_info, rpn_pre_nms_top_n=6000, rpn_post_nms_top_n=300,
                threshold=0.7, rpn_min_size=16, scales=(4., 8., 16., 32.), ratios=(0.5, 1., 2.),
                feature_stride=16, is_train=False, iou_loss=False):
    """Proposal_v3: cls_prob (B,2A,H,W), bbox_pred (B,4A,H,W), im_info (B,3) ->
    output (B,post,4), score (B,post,1)   (shapes proposal_v3-inl.h:196-214)."""
    _chk(cls_prob, "cls_prob", ndim=4)
    _chk(bbox_pred, "bbox_pred", ndim=4)
    _chk(im_info, "im_info", ndim=2)
    B, A2, H, W = cls_prob.shape
    A = A2 // 2
    if bbox_pred.shape != (B, 4 * A, H, W) or im_info.shape != (B, 3):
        raise ValueError("bbox_pred must be (B,4A,H,W) and im_info (B,3)")
    scales, ratios = list(scales), list(ratios)
    count = A * H * W
    pre = rpn_pre_nms_top_n if rpn_pre_nms_top_n > 0 else count
    pre = min(pre, count)
    post = int(rpn_post_nms_top_n) if not is_train else min(int(rpn_post_nms_top_n), pre)
    out = torch.empty((B, post, 4), device=cls_prob.device, dtype=torch.float32)
    score = torch.empty((B, post, 1), device=cls_prob.device, dtype=torch.float32)
    wsb = lib().cdll.sd_proposal_v3_workspace_bytes(B, A, H, W, int(rpn_pre_nms_top_n))
    ws = torch.empty(wsb, device=cls_prob.device, dtype=torch.uint8)
    lib().call("sd_proposal_v3_iou" if iou_loss else "sd_proposal_v3",
               _p(cls_prob), _p(bbox_pred), _p(im_info), _p(out), _p(score), B, A,
               H, W, int(rpn_pre_nms_top_n), int(rpn_post_nms_top_n), float(threshold),
               int(rpn_min_size), _farr(scales), len(scales), _farr(ratios), len(ratios),
               int(feature_stride), int(bool(is_train)), _p(ws), ctypes.c_size_t(wsb), _stream())
    return out, score


def get_top_proposal(bbox, score, top_n):
    """get_top_proposal CustomOp: bbox (B,N,4), score (B,N,1) -> (B,top_n,4), (B,top_n,1)."""
    _chk(bbox, "bbox", ndim=3)
    _chk(score, "score", ndim=3)
    B, N, _ = bbox.shape
    ob = torch.empty((B, int(top_n), 4), device=bbox.device, dtype=torch.float32)
    os_ = torch.empty((B, int(top_n), 1), device=bbox.device, dtype=torch.float32)
    lib().call("sd_get_top_proposal", _p(bbox), _p(score), B, N, int(top_n), _p(ob), _p(os_),
               _stream())
    return ob, os_


# --------------------------------------------------------------------------------------------------
# _contrib_DecodeBBox + test-time detection filter  (operator_cxx/contrib/decodebbox{-inl.h,.cc},
# detection_test.py:233-247) -- SURVEY 8(f) rank 2
# --------------------------------------------------------------------------------------------------
def decode_bbox(rois, bbox_pred, im_info, bbox_mean=(0., 0., 0., 0.), bbox_std=(.1, .1, .2, .2),
                class_agnostic=True, bbox_decode_type="xywh"):
    """DecodeBBox: rois (B,R,4), bbox_pred (B,R,4K), im_info (B,3) -> (B,R,4) if class_agnostic
    else (B,R,4K)  (decodebbox-inl.h:85-107)."""
    _chk(rois, "rois", ndim=3)
    _chk(bbox_pred, "bbox_pred", ndim=3)
    _chk(im_info, "im_info", ndim=2)
    if bbox_decode_type not in ("xywh", "xyxy"):
        raise ValueError("bbox_decode_type must be 'xywh' or 'xyxy'")
    B, R, _ = rois.shape
    K = bbox_pred.shape[2] // 4
    out = torch.empty((B, R, 4 if class_agnostic else 4 * K), device=rois.device,
                      dtype=torch.float32)
    lib().call("sd_decode_bbox", _p(rois), _p(bbox_pred), _p(im_info), _p(out), B, R, K,
               _farr(bbox_mean), _farr(bbox_std), int(bool(class_agnostic)),
               int(bbox_decode_type == "xyxy"), _stream())
    return out


def det_filter(bbox_xyxy, cls_score, min_det_score):
    """Per (image, class) rows with score > min_det_score as [box, score] (detection_test.py:
    236-247), in sd_soft_nms_batched's layout: dets (B*K,R,5), counts (B*K) int32."""
    _chk(bbox_xyxy, "bbox_xyxy", ndim=3)
    _chk(cls_score, "cls_score", ndim=3)
    B, R, K = cls_score.shape
    Kb = bbox_xyxy.shape[2] // 4
    dets = torch.empty((B * K, R, 5), device=cls_score.device, dtype=torch.float32)
    counts = torch.empty((B * K,), device=cls_score.device, dtype=torch.int32)
    lib().call("sd_det_filter", _p(bbox_xyxy), _p(cls_score), B, R, K, Kb, float(min_det_score),
               _p(dets), _p(counts), _stream())
    return dets, counts
