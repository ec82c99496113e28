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
