"""MXNet CustomOp adapter: the reference's operator names on top of the HIP C ABI.

The reference (tusen-ai/simpledet) reaches its detection ops by NAME from the symbol graph
(`mx.sym.contrib.ROIAlign_v2`, `mx.sym.ROIPooling_v1`, `mx.sym.ProposalTarget`,
`mx.sym.contrib.GenAnchor`, `mx.sym.contrib.NMS`, `mx.sym.contrib.DeformableConvolution`, and the
Python CustomOp `assign_layer_fpn`).  Its only run-time extension hook is `mx.operator.CustomOp`
(canonical example operator_py/bbox_target.py:96-219), so this module

  1. registers one CustomOp per operator under the name  sd_<reference op name>, with the
     reference's argument names, output names, NUMBER OF VISIBLE OUTPUTS and parameter names
     (roi_align_v2.cc:170-186, roi_pooling_v1-inl.h:144-237, proposal_target-inl.h:283-330,
     generate_anchor-inl.h:70-118, nms-inl.h:124-160), and
  2. `install()` aliases the reference names to `mx.sym.Custom(op_type=...)` so symbol/builder.py,
     models/ and config/ run unchanged.

`import mxnet` happens inside `register()`: importing this module never needs MXNet (it is absent
in the build container; tests drive the adapter through a tiny stub of the mx.operator interface).

Data path: NDArray -> raw device pointer through MXNet's C API (MXNDArrayGetData) -> C ABI on the
stream `install(stream=...)` names (default: the NULL stream) -> sd_stream_synchronize on that stream before
forward()/backward() return, because MXNet treats a CustomOp's outputs as written when the callback returns
(threading).

The ordering ASSUMPTION of the default (stated because no MXNet build is available here to check it): MXNet's
engine has made the inputs ready before it calls the callback (`wait_to_read` on every input is a no-op then, and is
issued anyway); the kernels are launched on the legacy NULL stream, which the HIP runtime orders against every
blocking stream of the device -- MXNet's per-device compute streams are created blocking
(`mshadow::Stream<gpu>`: hipStreamCreate, not hipStreamNonBlocking) -- and the host-side synchronise at the end
makes the outputs complete before the engine marks them written.  That serialises the device per op, which is
the price of the only run-time hook the reference has.  The library itself needs none of it: every entry point
takes the stream and neither synchronises nor touches another stream (tests/test_mxnet_plugin.py runs the adapter's
ops back to back on a side stream with the per-op synchronise switched off).  A host that can hand the adapter
the stream MXNet orders the op's outputs on -- `install(stream=<int | callable>, sync=False)` -- gets asynchronous
launches; the FCompute shim of INTEGRATION.md (B) is that design inside MXNet (`ctx.get_stream<gpu>()`,
roi_align_v2-inl.h:182).
"""
import ctypes
import os
from ast import literal_eval

from ._lib import SD_ERR_UNSUPPORTED, SimpleDetOpsError, lib

REQ = {"null": 0, "write": 1, "inplace": 2, "add": 3}
_PREFIX = "sd_"
_state = {"mx": None, "registered": False, "rng": {}, "stream": None, "sync": True}


# ------------------------------------------------------------------------------------ helpers ----
def _ptr(nd):
    """Raw device pointer of an mx.nd.NDArray (MXNDArrayGetData), or of any object that exposes
    `data_ptr()` (the test stub wraps torch tensors)."""
    if nd is None:
        return None
    if hasattr(nd, "data_ptr"):
        return ctypes.c_void_p(nd.data_ptr())
    mx = _state["mx"]
    p = ctypes.c_void_p()
    rc = mx.base._LIB.MXNDArrayGetData(nd.handle, ctypes.byref(p))
    if rc != 0:
        raise RuntimeError("MXNDArrayGetData failed")
    return p


def _wait(*arrs):
    for a in arrs:
        if a is not None and hasattr(a, "wait_to_read"):
            a.wait_to_read()


def _req(r):
    return REQ[r] if isinstance(r, str) else int(r)


def _stream():
    """hipStream_t the adapter launches on: install(stream=...) -- None = the NULL stream, an int / c_void_p, or
    a callable evaluated per call (e.g. `lambda: torch.cuda.current_stream().cuda_stream`)."""
    st = _state.get("stream")
    if callable(st):
        st = st()
    if st is None or isinstance(st, ctypes.c_void_p):
        return st
    return ctypes.c_void_p(int(st))


def _call(name, *args):
    """lib().call with the adapter's stream in place of a trailing None where the entry point's last parameter is
    `void* stream` (read off include/simpledet_ops.h)."""
    proto = lib().protos.get(name)
    if args and args[-1] is None and proto and proto[1] and proto[1][-1][0] == "stream":
        args = args[:-1] + (_stream(),)
    return lib().call(name, *args)


def _sync():
    if _state.get("sync", True):
        lib().call("sd_stream_synchronize", _stream())


def _tuple(v, n=None, typ=float):
    t = literal_eval(v) if isinstance(v, str) else v
    if not isinstance(t, (tuple, list)):
        t = (t,) * (n or 1)
    return tuple(typ(x) for x in t)


def _bool(v):
    if isinstance(v, str):
        return v.strip().lower() in ("1", "true", "yes")
    return bool(v)


def _iarr(vals):
    return (ctypes.c_int * len(vals))(*[int(v) for v in vals])


def _darr(vals):
    return (ctypes.c_double * len(vals))(*[float(v) for v in vals])


def _scratch(like, nbytes):
    """Device scratch owned by MXNet (an NDArray on the same context)."""
    mx = _state["mx"]
    return mx.nd.empty(((int(nbytes) + 3) // 4,), ctx=like.context, dtype="float32")


def _no_add(req):
    """forward outputs: kWriteTo / kWriteInplace / kNullOp only.  The kernels overwrite their
    outputs, so honouring kAddTo would need a temporary; no graph of the reference requests it
    (MXNet uses kAddTo for gradients), hence it is refused loudly instead of silently ignored."""
    for r in req:
        if _req(r) == REQ["add"]:
            raise RuntimeError("forward outputs do not support req='add' (kAddTo)")


def _require_write(req, names):
    for r, n in zip(req, names):
        if _req(r) not in (REQ["write"], REQ["null"]):
            raise RuntimeError("%s requires kWriteTo (got req=%s)" % (n, r))


# ---------------------------------------------------------------------------------- operators ----
def _build_ops(mx):
    CustomOp, CustomOpProp = mx.operator.CustomOp, mx.operator.CustomOpProp
    ops = {}

    # ---- _contrib_ROIAlign_v2: 2 inputs, 3 outputs (1 visible) ----
    class ROIAlignV2(CustomOp):
        def __init__(self, pooled_size, spatial_scale):
            super().__init__()
            self.ph, self.pw = pooled_size
            self.scale = spatial_scale

        def forward(self, is_train, req, in_data, out_data, aux):
            _no_add(req)
            data, rois = in_data
            _wait(data, rois)
            B, C, H, W = data.shape
            R = rois.shape[1]
            wsb = lib().cdll.sd_roi_align_v2_workspace_bytes(B, R)
            ws = _scratch(data, wsb)
            _call("sd_roi_align_v2_fwd_ws", _ptr(data), _ptr(rois), _ptr(out_data[0]),
                       _ptr(out_data[1]), _ptr(out_data[2]), B, C, H, W, R, self.ph, self.pw,
                       float(self.scale), _ptr(ws), ctypes.c_size_t(wsb), None)
            _sync()

        def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
            data, rois = in_data
            _wait(out_grad[0], rois, out_data[1], out_data[2])
            B, C, H, W = data.shape
            R = rois.shape[1]
            _call("sd_roi_align_v2_bwd", _ptr(out_grad[0]), _ptr(rois), _ptr(out_data[1]),
                       _ptr(out_data[2]), _ptr(in_grad[0]), _ptr(in_grad[1]), _req(req[0]),
                       _req(req[1]), B, C, H, W, R, self.ph, self.pw, float(self.scale), None)
            _sync()

    class ROIAlignV2Prop(CustomOpProp):
        def __init__(self, pooled_size, spatial_scale):
            super().__init__(need_top_grad=True)
            self.pooled_size = _tuple(pooled_size, 2, int)
            self.spatial_scale = float(spatial_scale)
            if self.pooled_size[0] <= 0 or self.pooled_size[1] <= 0:
                raise ValueError("ROIAlignParam: pooled_size must be nonzero")
            if not 0.0 <= self.spatial_scale <= 1.0:
                raise ValueError("spatial_scale must be in [0, 1]")

        def list_arguments(self):
            return ["data", "rois"]

        def list_outputs(self):
            return ["output", "maxidx_x", "maxidx_y"]

        num_visible_outputs = 1

        def infer_shape(self, in_shape):
            d, b = in_shape
            if len(d) != 4:
                raise ValueError("data should be a 4D tensor")
            if len(b) != 3 or b[2] != 4:
                raise ValueError("bbox should be a 3D tensor of shape [batch, rois, 4]")
            o = (b[0], b[1], d[1], self.pooled_size[0], self.pooled_size[1])
            return [d, b], [o, o, o]

        def create_operator(self, ctx, shapes, dtypes):
            return ROIAlignV2(self.pooled_size, self.spatial_scale)

        def declare_backward_dependency(self, out_grad, in_data, out_data):
            # ROIAlignGrad_v2 (roi_align_v2-inl.h:206-218): dY, rois, maxidx_x, maxidx_y
            return [out_grad[0], in_data[1], out_data[1], out_data[2]]

    ops["_contrib_ROIAlign_v2"] = (ROIAlignV2Prop, ("contrib", "ROIAlign_v2"))

    # ---- ROIPooling_v1: 2 inputs, 2 outputs (1 visible) ----
    class ROIPoolingV1(CustomOp):
        def __init__(self, pooled_size, spatial_scale):
            super().__init__()
            self.ph, self.pw = pooled_size
            self.scale = spatial_scale

        def forward(self, is_train, req, in_data, out_data, aux):
            _no_add(req)
            data, rois = in_data
            _wait(data, rois)
            B, C, H, W = data.shape
            _call("sd_roi_pool_v1_fwd", _ptr(data), _ptr(rois), _ptr(out_data[0]),
                       _ptr(out_data[1]), B, C, H, W, rois.shape[0], self.ph, self.pw,
                       float(self.scale), None)
            _sync()

        def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
            data, rois = in_data
            _wait(out_grad[0], rois, out_data[1])
            B, C, H, W = data.shape
            _call("sd_roi_pool_v1_bwd", _ptr(out_grad[0]), _ptr(rois), _ptr(out_data[1]),
                       _ptr(in_grad[0]), _ptr(in_grad[1]), _req(req[0]), _req(req[1]), B, C, H, W,
                       rois.shape[0], self.ph, self.pw, float(self.scale), None)
            _sync()

    class ROIPoolingV1Prop(CustomOpProp):
        def __init__(self, pooled_size, spatial_scale):
            super().__init__(need_top_grad=True)
            self.pooled_size = _tuple(pooled_size, 2, int)
            self.spatial_scale = float(spatial_scale)

        def list_arguments(self):
            return ["data", "rois"]

        def list_outputs(self):
            return ["output", "maxidx"]

        num_visible_outputs = 1

        def infer_shape(self, in_shape):
            d, b = in_shape
            if len(d) != 4:
                raise ValueError("data should be a 4D tensor")
            if len(b) != 2 or b[1] != 5:
                raise ValueError("bbox should be a 2D tensor of shape [batch, 5]")
            o = (b[0], d[1], self.pooled_size[0], self.pooled_size[1])
            return [d, b], [o, o]

        def create_operator(self, ctx, shapes, dtypes):
            return ROIPoolingV1(self.pooled_size, self.spatial_scale)

        def declare_backward_dependency(self, out_grad, in_data, out_data):
            return [out_grad[0], in_data[1], out_data[1]]

    ops["ROIPooling_v1"] = (ROIPoolingV1Prop, (None, "ROIPooling_v1"))

    # ---- ProposalTarget: 2 inputs, 5 outputs (4 visible unless output_iou) ----
    class ProposalTarget(CustomOp):
        def __init__(self, p):
            super().__init__()
            self.p = p

        def forward(self, is_train, req, in_data, out_data, aux):
            _no_add(req)
            from .ops import ProposalTargetParam
            _require_write(req[:4], ["roi_output", "label", "bbox_target", "bbox_weight"])
            rois, gt = in_data[0], in_data[1]
            vr = in_data[2] if len(in_data) > 2 else None  # ProposalTarget_v2: valid_ranges
            _wait(*in_data)
            p = self.p
            B = p["batch_images"]
            N = int(_numel(rois.shape) // (B * 4))
            M = int(_numel(gt.shape) // (B * 5))
            cp = ProposalTargetParam()
            cp.num_classes, cp.batch_images, cp.image_rois = p["num_classes"], B, p["image_rois"]
            cp.fg_fraction, cp.fg_thresh = p["fg_fraction"], p["fg_thresh"]
            cp.bg_thresh_hi, cp.bg_thresh_lo = p["bg_thresh_hi"], p["bg_thresh_lo"]
            cp.proposal_without_gt, cp.class_agnostic = int(p["proposal_without_gt"]), int(p["class_agnostic"])
            for i in range(4):
                cp.bbox_mean[i], cp.bbox_std[i], cp.bbox_weight[i] = (p["bbox_mean"][i], p["bbox_std"][i],
                                                                       p["bbox_weight"][i])
            key = str(rois.context)
            if key not in _state["rng"]:
                # libc's global rand() state of a process that never called srand (seed 1): one
                # stream per device, shared by every ProposalTarget node like the libc global is
                host = (ctypes.c_int32 * 33)()
                _call("sd_glibc_srand_host", ctypes.c_uint32(1), host)
                _state["rng"][key] = _state["mx"].nd.array(list(host), ctx=rois.context, dtype="int32")
            rng = _state["rng"][key]
            wsb = lib().cdll.sd_proposal_target_workspace_bytes(B, N, M)
            ws = _scratch(rois, wsb)
            if vr is not None:
                _call("sd_proposal_target_v2", _ptr(rois), _ptr(gt), _ptr(vr),
                           int(p.get("filter_scales", False)), N, M, ctypes.byref(cp), _ptr(rng),
                           _ptr(out_data[0]), _ptr(out_data[1]), _ptr(out_data[2]), _ptr(out_data[3]),
                           _ptr(out_data[4]), None, _ptr(ws), ctypes.c_size_t(wsb), None)
            else:
                _call("sd_proposal_target", _ptr(rois), _ptr(gt), N, M, ctypes.byref(cp), _ptr(rng),
                           _ptr(out_data[0]), _ptr(out_data[1]), _ptr(out_data[2]), _ptr(out_data[3]),
                           _ptr(out_data[4]), None, _ptr(ws), ctypes.c_size_t(wsb), None)
            _sync()

        def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
            # proposal_target-inl.h:272-276 / proposal_target_v2-inl.h:298-311: input gradients are zero
            for i in range(len(in_grad)):
                self.assign(in_grad[i], req[i], 0)

    def _numel(shape):
        n = 1
        for s in shape:
            n *= int(s)
        return n

    class ProposalTargetProp(CustomOpProp):
        def __init__(self, num_classes, batch_images, image_rois, fg_thresh, bg_thresh_hi,
                     bg_thresh_lo, fg_fraction="0.25", proposal_without_gt="False",
                     class_agnostic="False", output_iou="False", bbox_mean="(0,0,0,0)",
                     bbox_std="(0.1,0.1,0.2,0.2)", bbox_weight="(1,1,1,1)"):
            super().__init__(need_top_grad=False)
            self.p = dict(num_classes=int(num_classes), batch_images=int(batch_images),
                          image_rois=int(image_rois), fg_thresh=float(fg_thresh),
                          bg_thresh_hi=float(bg_thresh_hi), bg_thresh_lo=float(bg_thresh_lo),
                          fg_fraction=float(fg_fraction),
                          proposal_without_gt=_bool(proposal_without_gt),
                          class_agnostic=_bool(class_agnostic), output_iou=_bool(output_iou),
                          bbox_mean=_tuple(bbox_mean, 4), bbox_std=_tuple(bbox_std, 4),
                          bbox_weight=_tuple(bbox_weight, 4))
            self.num_visible_outputs = 5 if self.p["output_iou"] else 4

        def list_arguments(self):
            return ["rois", "gt_boxes"]

        def list_outputs(self):
            return ["roi_output", "label", "bbox_target", "bbox_weight", "match_gt_iou"]

        def infer_shape(self, in_shape):
            p = self.p
            B, S, K = p["batch_images"], p["image_rois"], p["num_classes"]
            return in_shape, [(B, S, 4), (B, S), (B, S, K * 4), (B, S, K * 4), (B, S)]

        def create_operator(self, ctx, shapes, dtypes):
            return ProposalTarget(self.p)

        def declare_backward_dependency(self, out_grad, in_data, out_data):
            return []

    ops["ProposalTarget"] = (ProposalTargetProp, (None, "ProposalTarget"))

    # ---- ProposalTarget_v2 (proposal_target_v2-inl.h): + valid_ranges input, filter_scales ----
    class ProposalTargetV2Prop(ProposalTargetProp):
        def __init__(self, num_classes, batch_images, image_rois, fg_thresh, bg_thresh_hi,
                     bg_thresh_lo, proposal_without_gt, fg_fraction="0.25", class_agnostic="False",
                     ohem="False", output_iou="False", bbox_mean="(0,0,0,0)",
                     bbox_std="(0.1,0.1,0.2,0.2)", bbox_weight="(1,1,1,1)", filter_scales="False"):
            super().__init__(num_classes, batch_images, image_rois, fg_thresh, bg_thresh_hi,
                             bg_thresh_lo, fg_fraction, proposal_without_gt, class_agnostic,
                             output_iou, bbox_mean, bbox_std, bbox_weight)
            if _bool(ohem):
                raise ValueError("ProposalTarget_v2: OHEM not Implemented.")  # as the reference (:216-217)
            if self.p["image_rois"] < 0:
                raise ValueError("ProposalTarget_v2: image_rois=-1 is undefined in the reference")
            self.p["filter_scales"] = _bool(filter_scales)

        def list_arguments(self):
            return ["rois", "gt_boxes", "valid_ranges"]

    ops["ProposalTarget_v2"] = (ProposalTargetV2Prop, (None, "ProposalTarget_v2"))

    # ---- ProposalMaskTarget (proposal_mask_target-inl.h): rois, gt_boxes, gt_polys [, valid_ranges]
    #      -> the five ProposalTarget outputs + mask_target (+ mask_ratio with output_ratio) ----
    class ProposalMaskTarget(CustomOp):
        def __init__(self, p):
            super().__init__()
            self.p = p

        def forward(self, is_train, req, in_data, out_data, aux):
            _no_add(req)
            from .ops import ProposalTargetParam
            _require_write(req[:6], ["roi_output", "label", "bbox_target", "bbox_weight", "match_gt_iou",
                                     "mask_target"])
            rois, gt, polys = in_data[0], in_data[1], in_data[2]
            vr = in_data[3] if len(in_data) > 3 else None
            _wait(*in_data)
            p = self.p
            B = p["batch_images"]
            N = int(_numel(rois.shape) // (B * 4))
            M = int(_numel(gt.shape) // (B * 5))
            L = int(polys.shape[2])
            cp = ProposalTargetParam()
            cp.num_classes, cp.batch_images, cp.image_rois = p["num_classes"], B, p["image_rois"]
            cp.fg_fraction, cp.fg_thresh = p["fg_fraction"], p["fg_thresh"]
            cp.bg_thresh_hi, cp.bg_thresh_lo = p["bg_thresh_hi"], p["bg_thresh_lo"]
            cp.proposal_without_gt, cp.class_agnostic = int(p["proposal_without_gt"]), int(p["class_agnostic"])
            for i in range(4):
                cp.bbox_mean[i], cp.bbox_std[i], cp.bbox_weight[i] = (p["bbox_mean"][i], p["bbox_std"][i],
                                                                       p["bbox_weight"][i])
            key = str(rois.context)
            if key not in _state["rng"]:
                host = (ctypes.c_int32 * 33)()
                _call("sd_glibc_srand_host", ctypes.c_uint32(1), host)
                _state["rng"][key] = _state["mx"].nd.array(list(host), ctx=rois.context, dtype="int32")
            rng = _state["rng"][key]
            if p["output_ratio"]:  # proposal_mask_target-inl.h:159-161: the seventh output, kWriteTo
                _require_write(req[6:7], ["mask_ratio"])
                mp = p["max_raster_pixels"]
                wsb = lib().cdll.sd_proposal_mask_target_ratio_workspace_bytes(
                    B, N, M, p["image_rois"], ctypes.c_float(p["fg_fraction"]), mp)
                ws = _scratch(rois, wsb)
                _call("sd_proposal_mask_target_ratio", _ptr(rois), _ptr(gt), _ptr(polys), _ptr(vr),
                           int(p["filter_scales"]), N, M, L, p["mask_size"], ctypes.byref(cp), _ptr(rng),
                           _ptr(out_data[0]), _ptr(out_data[1]), _ptr(out_data[2]), _ptr(out_data[3]),
                           _ptr(out_data[4]), _ptr(out_data[5]), _ptr(out_data[6]), mp, None, _ptr(ws),
                           ctypes.c_size_t(wsb), None)
                _sync()
                return
            wsb = lib().cdll.sd_proposal_target_workspace_bytes(B, N, M)
            ws = _scratch(rois, wsb)
            _call("sd_proposal_mask_target", _ptr(rois), _ptr(gt), _ptr(polys), _ptr(vr),
                       int(p["filter_scales"]), N, M, L, p["mask_size"], ctypes.byref(cp), _ptr(rng),
                       _ptr(out_data[0]), _ptr(out_data[1]), _ptr(out_data[2]), _ptr(out_data[3]),
                       _ptr(out_data[4]), _ptr(out_data[5]), None, _ptr(ws), ctypes.c_size_t(wsb), None)
            _sync()

        def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
            for i in range(len(in_grad)):
                self.assign(in_grad[i], req[i], 0)

    class ProposalMaskTargetProp(ProposalTargetProp):
        def __init__(self, num_classes, batch_images, image_rois, mask_size, fg_thresh,
                     bg_thresh_hi, bg_thresh_lo, proposal_without_gt, fg_fraction="0.25",
                     class_agnostic="False", ohem="False", output_ratio="False", output_iou="False",
                     filter_scales="False", bbox_mean="(0,0,0,0)", bbox_std="(0.1,0.1,0.2,0.2)",
                     bbox_weight="(1,1,1,1)", num_args=None, max_raster_pixels="1982464"):
            # num_args is the reference op's key_var_num_args (proposal_mask_target.cc:485): MXNet's
            # front end fills it in from the number of symbol inputs and no call site passes it
            # (models/maskrcnn/builder.py:115,184), so it defaults to what filter_scales implies
            super().__init__(num_classes, batch_images, image_rois, fg_thresh, bg_thresh_hi,
                             bg_thresh_lo, fg_fraction, proposal_without_gt, class_agnostic,
                             output_iou, bbox_mean, bbox_std, bbox_weight)
            if _bool(ohem):
                raise ValueError("ProposalMaskTarget: OHEM not Implemented.")
            # output_ratio (mask scoring R-CNN, models/msrcnn/builder.py:219-237): the seventh output.
            # max_raster_pixels is this adapter's own attribute (not the reference's): the bound on
            # the image-resolution rasters the ratio counts, default 1408 x 1408
            self.p["output_ratio"] = _bool(output_ratio)
            self.p["max_raster_pixels"] = int(max_raster_pixels)
            if self.p["image_rois"] < 0:
                raise ValueError("ProposalMaskTarget: image_rois=-1 is undefined in the reference")
            self.p["filter_scales"] = _bool(filter_scales)
            self.p["mask_size"] = int(mask_size)
            want = 4 if self.p["filter_scales"] else 3
            self.num_args = want if num_args is None else int(num_args)
            if self.num_args != want:
                raise ValueError("num_args=%d but filter_scales=%s takes %d inputs"
                                 % (self.num_args, self.p["filter_scales"], want))
            # proposal_mask_target-inl.h:387-409 (the graph unpacks match_gt_iou either way:
            # models/maskrcnn/builder.py:115, models/msrcnn/builder.py:219)
            self.num_visible_outputs = 7 if self.p["output_ratio"] else 6

        def list_arguments(self):
            base = ["rois", "gt_boxes", "gt_polys"]
            return base + ["valid_ranges"] if self.p["filter_scales"] else base

        def list_outputs(self):
            base = ["roi_output", "label", "bbox_target", "bbox_weight", "match_gt_iou", "mask_target"]
            return base + ["mask_ratio"] if self.p["output_ratio"] else base

        def infer_shape(self, in_shape):
            p = self.p
            B, S, K = p["batch_images"], p["image_rois"], p["num_classes"]
            import numpy as np
            FG = int(np.float32(S) * np.float32(p["fg_fraction"]))
            out = [(B, S, 4), (B, S), (B, S, K * 4), (B, S, K * 4), (B, S),
                   (B, FG, p["mask_size"], p["mask_size"])]
            if p["output_ratio"]:
                out.append((B, FG))  # -inl.h:453-456
            return in_shape, out

        def create_operator(self, ctx, shapes, dtypes):
            return ProposalMaskTarget(self.p)

    ops["ProposalMaskTarget"] = (ProposalMaskTargetProp, (None, "ProposalMaskTarget"))

    # ---- _contrib_GenAnchor: 1 input (shape only), 1 output ----
    class GenAnchor(CustomOp):
        def __init__(self, scales, ratios, stride):
            super().__init__()
            self.scales, self.ratios, self.stride = scales, ratios, stride

        def forward(self, is_train, req, in_data, out_data, aux):
            _no_add(req)
            _require_write(req[:1], ["output"])
            H, W = in_data[0].shape[2], in_data[0].shape[3]
            _call("sd_gen_anchor", _ptr(out_data[0]), H, W, self.stride, _darr(self.scales),
                       len(self.scales), _darr(self.ratios), len(self.ratios), None)
            _sync()

        def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
            self.assign(in_grad[0], req[0], 0)

    class GenAnchorProp(CustomOpProp):
        def __init__(self, scales="(4,8,16,32)", ratios="(0.5,1,2)", feature_stride="16"):
            super().__init__(need_top_grad=False)
            self.scales, self.ratios = _tuple(scales), _tuple(ratios)
            self.stride = int(feature_stride)

        def list_arguments(self):
            return ["cls_prob"]

        def list_outputs(self):
            return ["output"]

        def infer_shape(self, in_shape):
            d = in_shape[0]
            if len(d) != 4:
                raise ValueError("cls_prob should be a 4D tensor")
            A = len(self.scales) * len(self.ratios)
            return in_shape, [(d[2] * d[3] * A, 4)]

        def create_operator(self, ctx, shapes, dtypes):
            return GenAnchor(self.scales, self.ratios, self.stride)

        def declare_backward_dependency(self, out_grad, in_data, out_data):
            return []

    ops["_contrib_GenAnchor"] = (GenAnchorProp, ("contrib", "GenAnchor"))

    # ---- _contrib_NMS: 1 input, 2 outputs (score visible only with output_score) ----
    class NMS(CustomOp):
        def __init__(self, pre, post, thr, already_sorted):
            super().__init__()
            self.pre, self.post, self.thr, self.sorted = pre, post, thr, already_sorted

        def forward(self, is_train, req, in_data, out_data, aux):
            _no_add(req)
            rois = in_data[0]
            _wait(rois)
            B, N, _ = rois.shape
            wsb = lib().cdll.sd_nms_workspace_bytes(B, N, self.pre)
            ws = _scratch(rois, wsb)
            pre = min(self.pre if self.pre > 0 else N, N)
            if self.post > pre:
                # NMSProp::InferShape always declares (B, post, .) but the op writes min(post, pre)
                # rows per image, image i at row offset i*min(post, pre) of the flat buffer
                # (nms.cu:277,352-354), and never touches the tail.  Same bytes here; the tail
                # (uninitialised in the reference) is zeroed.
                self.assign(out_data[0], "write", 0)
                self.assign(out_data[1], "write", 0)
            _call("sd_nms", _ptr(rois), B, N, self.pre, self.post, float(self.thr), 0,
                       int(self.sorted), _ptr(out_data[0]), _ptr(out_data[1]), None, _ptr(ws),
                       ctypes.c_size_t(wsb), None)
            _sync()

        def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
            self.assign(in_grad[0], req[0], 0)  # nms.cu:379-381

    class NMSProp(CustomOpProp):
        def __init__(self, rpn_pre_nms_top_n="6000", rpn_post_nms_top_n="300", threshold="0.7",
                     output_score="False", already_sorted="False", workspace="256"):
            super().__init__(need_top_grad=False)
            self.pre, self.post = int(rpn_pre_nms_top_n), int(rpn_post_nms_top_n)
            self.thr = float(threshold)
            self.sorted = _bool(already_sorted)
            self.num_visible_outputs = 2 if _bool(output_score) else 1

        def list_arguments(self):
            return ["rois"]

        def list_outputs(self):
            return ["output", "score"]

        def infer_shape(self, in_shape):
            d = in_shape[0]
            if len(d) != 3 or d[2] != 5:
                raise ValueError("Input:[bbox] must be (batch, rois, 5)")
            return in_shape, [(d[0], self.post, 4), (d[0], self.post, 1)]

        def create_operator(self, ctx, shapes, dtypes):
            return NMS(self.pre, self.post, self.thr, self.sorted)

        def declare_backward_dependency(self, out_grad, in_data, out_data):
            return []

    ops["_contrib_NMS"] = (NMSProp, ("contrib", "NMS"))

    # ---- assign_layer_fpn (models/FPN/assign_layer_fpn.py): 1 input, len(rcnn_stride) outputs ----
    class AssignLayerFPN(CustomOp):
        def __init__(self, strides, scale0, lvl0):
            super().__init__()
            self.strides, self.scale0, self.lvl0 = strides, scale0, lvl0

        def forward(self, is_train, req, in_data, out_data, aux):
            _no_add(req)
            rois = in_data[0]
            _wait(rois)
            n = 1
            for s in rois.shape[:-1]:
                n *= int(s)
            mx_ = _state["mx"]
            per = mx_.nd.empty((len(self.strides),) + tuple(rois.shape), ctx=rois.context)
            _call("sd_fpn_roi_assign", _ptr(rois), n, _iarr(self.strides), len(self.strides),
                       float(self.scale0), float(self.lvl0), _ptr(per), None, None)
            _sync()
            for i in range(len(self.strides)):
                self.assign(out_data[i], req[i], per[i])

        def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
            self.assign(in_grad[0], req[0], 0)

    class AssignLayerFPNProp(CustomOpProp):
        def __init__(self, rcnn_stride, roi_canonical_scale, roi_canonical_level):
            super().__init__(need_top_grad=False)
            self.rcnn_stride = _tuple(rcnn_stride, typ=int)
            self.roi_canonical_scale = int(roi_canonical_scale)
            self.roi_canonical_level = int(roi_canonical_level)

        def list_arguments(self):
            return ["rois"]

        def list_outputs(self):
            return ["rois_s{}".format(s) for s in self.rcnn_stride]

        def infer_shape(self, in_shape):
            return [in_shape[0]], [in_shape[0]] * len(self.rcnn_stride)

        def create_operator(self, ctx, shapes, dtypes):
            return AssignLayerFPN(self.rcnn_stride, self.roi_canonical_scale,
                                  self.roi_canonical_level)

        def declare_backward_dependency(self, out_grad, in_data, out_data):
            return []

    ops["assign_layer_fpn"] = (AssignLayerFPNProp, None)

    # ---- _contrib_DeformableConvolution: data, offset, weight[, bias] -> output.  Upstream MXNet 1.6
    #      DeformableConvolutionParam: kernel, stride, dilate, pad, num_filter, num_group,
    #      num_deformable_group, workspace, no_bias (default False), layout.  Call sites:
    #      models/dcn/builder.py:14-17 (no_bias=True), models/RepPoints/builder.py:215-245 (bias),
    #      models/sepc/sepc_dconv.py:12-16 (num_group / bias passed through),
    #      models/tridentnet/resnet_v1.py:85-90 (weight / bias shared between branches) ----
    class DeformConv(CustomOp):
        def __init__(self, g):
            super().__init__()
            self.g = g

        def _ws(self, x):
            g = self.g
            N, C, H, W = x.shape
            n = lib().cdll.sd_deform_conv_workspace_bytes(N, C, H, W, g["kh"], g["kw"], g["pad"],
                                                          g["stride"], g["dil"])
            return _scratch(x, n), n

        def forward(self, is_train, req, in_data, out_data, aux):
            _no_add(req)
            x, off, w = in_data[:3]
            b = in_data[3] if self.g["bias"] else None
            _wait(*in_data)
            g = self.g
            N, C, H, W = x.shape
            # training with cache_col: im2col + GEMM, the col matrix stays alive until this node's backward,
            # which then skips its own im2col (620 MB per layer at the baseline; cache_col="False" trades it
            # for the fused col-free forward and a backward that recomputes col).  Otherwise no col matrix:
            # deformable sampling fused into the GEMM (a few MB of workspace instead of N*C*9*Ho*Wo*4 bytes;
            # shapes the fused kernel does not take run im2col + GEMM behind the same entry point)
            keep = 1 if (is_train and g["cache_col"]) else 0
            n = int(lib().cdll.sd_deform_convolution_fwd_workspace_bytes(
                N, C, H, W, g["F"], g["kh"], g["kw"], g["pad"], g["stride"], g["dil"], g["dg"], g["G"], keep))
            ws = _scratch(x, n)
            _call("sd_deform_convolution_fwd", _ptr(x), _ptr(off), _ptr(w), _ptr(b), _ptr(out_data[0]), N, C,
                       H, W, g["F"], g["kh"], g["kw"], g["pad"], g["stride"], g["dil"], g["dg"], g["G"], keep,
                       _ptr(ws), ctypes.c_size_t(n), None)
            self._fwd_ws = (ws, tuple(x.shape)) if keep else None
            _sync()

        def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
            x, off, w = in_data[:3]
            _wait(out_grad[0], *in_data)
            g = self.g
            N, C, H, W = x.shape
            ws, n = self._ws(x)
            kept = getattr(self, "_fwd_ws", None)
            self._fwd_ws = None
            col = None
            if kept is not None and kept[1] == tuple(x.shape):
                col = ctypes.c_void_p(lib().cdll.sd_deform_conv_col_of_workspace(_ptr(kept[0])))
            has_b = g["bias"]
            _call("sd_deform_convolution_bwd", _ptr(out_grad[0]), _ptr(x), _ptr(off), _ptr(w), col,
                       _ptr(in_grad[0]), _ptr(in_grad[1]), _ptr(in_grad[2]), _ptr(in_grad[3]) if has_b else None,
                       _req(req[0]), _req(req[1]), _req(req[2]), _req(req[3]) if has_b else REQ["null"],
                       N, C, H, W, g["F"], g["kh"], g["kw"], g["pad"], g["stride"], g["dil"], g["dg"], g["G"],
                       _ptr(ws), ctypes.c_size_t(n), None)
            _sync()

    class DeformConvProp(CustomOpProp):
        # the op's own parameter names (+ cache_col, this adapter's attribute); anything else, and any
        # value the kernels do not take, makes install()'s alias hand the call back to the constructor it
        # replaced (sd_supports) -- the graph then holds the native operator for that node
        PARAMS = ("kernel", "num_filter", "stride", "dilate", "pad", "num_group", "num_deformable_group", "no_bias",
                  "workspace", "layout", "cache_col")

        def __init__(self, kernel, num_filter, stride="(1,1)", dilate="(1,1)", pad="(0,0)",
                     num_group="1", num_deformable_group="1", no_bias="False", workspace="1024",
                     layout="None", cache_col="True"):
            # cache_col is this adapter's own attribute: keep the forward's col matrix for the
            # backward of the same node (training only)
            super().__init__(need_top_grad=True)
            why = self.sd_supports(dict(kernel=kernel, num_filter=num_filter, stride=stride, dilate=dilate, pad=pad,
                                        num_group=num_group, num_deformable_group=num_deformable_group,
                                        layout=layout))
            if why:
                raise ValueError("DeformableConvolution: " + why)
            k, s, d, p = (_tuple(kernel, 2, int), _tuple(stride, 2, int), _tuple(dilate, 2, int),
                          _tuple(pad, 2, int))
            self.g = dict(kh=k[0], kw=k[1], stride=s[0], dil=d[0], pad=p[0], F=int(num_filter),
                          dg=int(num_deformable_group), G=int(num_group), bias=not _bool(no_bias),
                          # SIMPLEDET_AMD_DCN_CACHE_COL=0: process-wide off switch (no node keeps 620 MB
                          # between its forward and backward, whatever its attribute says)
                          cache_col=_bool(cache_col) and os.environ.get("SIMPLEDET_AMD_DCN_CACHE_COL", "1") != "0")

        @classmethod
        def sd_supports(cls, params):
            """'' when the kernels take this parameter set, else the reason (install()'s alias then falls
            back to the native constructor).  params: str-valued, as MXNet hands them to a CustomOpProp."""
            for k in params:
                if k not in cls.PARAMS:
                    return "parameter %r is not one this operator takes" % k
            try:
                k = _tuple(params.get("kernel", "(0,0)"), 2, int)
                s, d, p = (_tuple(params.get(n, dflt), 2, int) for n, dflt in
                           (("stride", "(1,1)"), ("dilate", "(1,1)"), ("pad", "(0,0)")))
                int(params.get("num_filter", "0")), int(params.get("num_group", "1"))
                int(params.get("num_deformable_group", "1"))
            except Exception as e:
                return "unparsable parameter (%s)" % e
            if len(k) != 2 or len(s) != 2 or len(d) != 2 or len(p) != 2:
                return "2-D convolutions only"
            if s[0] != s[1] or d[0] != d[1] or p[0] != p[1]:
                return "square stride/dilate/pad only"
            if str(params.get("layout", "None")) not in ("None", "NCHW"):
                return "layout NCHW only"
            if int(params.get("num_group", "1")) < 1 or int(params.get("num_deformable_group", "1")) < 1:
                return "num_group / num_deformable_group must be positive"
            return ""

        def list_arguments(self):
            return ["data", "offset", "weight"] + (["bias"] if self.g["bias"] else [])

        def list_outputs(self):
            return ["output"]

        def infer_shape(self, in_shape):
            g = self.g
            d = in_shape[0]
            if len(d) != 4:
                raise ValueError("Input data should be 4D in batch-num_filter-y-x")
            if d[1] % g["G"] or g["F"] % g["G"]:
                raise ValueError("input / output num_filter must divide group size")
            if d[1] % g["dg"]:
                raise ValueError("input num_filter must divide deformable group size")
            Ho = (d[2] + 2 * g["pad"] - (g["dil"] * (g["kh"] - 1) + 1)) // g["stride"] + 1
            Wo = (d[3] + 2 * g["pad"] - (g["dil"] * (g["kw"] - 1) + 1)) // g["stride"] + 1
            off = (d[0], g["dg"] * 2 * g["kh"] * g["kw"], Ho, Wo)
            w = (g["F"], d[1] // g["G"], g["kh"], g["kw"])
            ins = [d, off, w] + ([(g["F"],)] if g["bias"] else [])
            return ins, [(d[0], g["F"], Ho, Wo)]

        def create_operator(self, ctx, shapes, dtypes):
            return DeformConv(self.g)

        def declare_backward_dependency(self, out_grad, in_data, out_data):
            # (the bias itself is not needed by the backward: deformable_convolution-inl.h Backward reads
            # out_grad, data, offset, weight)
            return [out_grad[0], in_data[0], in_data[1], in_data[2]]

    ops["_contrib_DeformableConvolution"] = (DeformConvProp, ("contrib", "DeformableConvolution"))

    def _fpn_packed(pooled):
        return tuple(pooled) in ((7, 7), (14, 14))

    # ---- fpn_roi_align: the whole FPNRoiAlign.get_roi_feature subgraph (models/FPN/builder.py:
    #      567-610: assign -> per level ROIAlign_v2 -> add_n) as ONE op: feats..., rois -> output ----
    class FPNRoIAlign(CustomOp):
        """outputs: output, then the op's private forward -> backward state.  7x7 / 14x14 pooling:
        one-byte arg-max + the per-RoI coordinate / tap table (sd_fpn_roi_align_fwd_packed);
        other sizes: the reference's two fp32 arg-max planes."""

        def __init__(self, strides, pooled, scale0, lvl0, fp16=False):
            super().__init__()
            self.strides, self.pooled, self.scale0, self.lvl0 = strides, pooled, scale0, lvl0
            self.packed = _fpn_packed(pooled)
            self.fp16 = fp16  # fp16 feature maps in, fp16 output out (packed pooling sizes only)

        def _levels(self, feats):
            ptrs = (ctypes.c_void_p * len(feats))(*[_ptr(f).value for f in feats])
            return ptrs, _iarr([f.shape[2] for f in feats]), _iarr([f.shape[3] for f in feats])

        def forward(self, is_train, req, in_data, out_data, aux):
            _no_add(req)
            feats, rois = in_data[:-1], in_data[-1]
            _wait(*in_data)
            B, C = feats[0].shape[:2]
            ptrs, Hs, Ws = self._levels(feats)
            wsb = lib().cdll.sd_fpn_roi_align_workspace_bytes(B, rois.shape[1])
            ws = _scratch(rois, wsb)
            fn = "sd_fpn_roi_align_fwd_packed" if self.packed else "sd_fpn_roi_align_fwd"

            def run(name, level_ptrs, out0):
                _call(name, level_ptrs, Hs, Ws, _iarr(self.strides), len(feats), _ptr(rois),
                           _ptr(out0), _ptr(out_data[1]), _ptr(out_data[2]), B, C, rois.shape[1],
                           self.pooled[0], self.pooled[1], float(self.scale0), float(self.lvl0),
                           _ptr(ws), ctypes.c_size_t(wsb), None)

            if not self.fp16 and self.packed and is_train:
                # training: ONE rois-only pre-pass for the step -- the backward's band lists / tap
                # tables are built here, into the op's fourth (private) output
                plan = out_data[3]
                _call(fn + "_plan", ptrs, Hs, Ws, _iarr(self.strides), len(feats), _ptr(rois),
                           _ptr(out_data[0]), _ptr(out_data[1]), _ptr(out_data[2]), B, C, rois.shape[1],
                           self.pooled[0], self.pooled[1], float(self.scale0), float(self.lvl0),
                           _ptr(ws), ctypes.c_size_t(wsb), _ptr(plan), ctypes.c_size_t(plan.size), None)
                self._planned = True
            elif not self.fp16:
                self._planned = False
                run(fn, ptrs, out_data[0])
            else:
                try:
                    run(fn + "_f16", ptrs, out_data[0])
                except SimpleDetOpsError as e:
                    if e.code != SD_ERR_UNSUPPORTED:
                        raise
                    # a shape the band-resident kernel does not take: the casts the reference graph
                    # carries itself (models/FPN/builder.py:581-586, 607-608) around the fp32 op
                    f32 = [_scratch(rois, f.size * 4).reshape(f.shape) for f in feats]
                    for f, g in zip(feats, f32):
                        _call("sd_cast_f16_to_f32", _ptr(f), _ptr(g), ctypes.c_size_t(f.size), None)
                    o32 = _scratch(rois, out_data[0].size * 4).reshape(out_data[0].shape)
                    p32 = (ctypes.c_void_p * len(f32))(*[_ptr(g).value for g in f32])
                    run(fn, p32, o32)
                    _call("sd_cast_f32_to_f16", _ptr(o32), _ptr(out_data[0]),
                               ctypes.c_size_t(o32.size), REQ["write"], None)
            _sync()

        def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
            feats, rois = in_data[:-1], in_data[-1]
            _wait(out_grad[0], rois, *out_data[1:])
            rq = {_req(r) for r in req[:-1]}
            if len(rq) != 1:
                raise RuntimeError("fpn_roi_align: all feature gradients must share one req")
            B, C = feats[0].shape[:2]
            req_data = rq.pop()
            og, grads16 = out_grad[0], None
            if self.fp16 and self.packed:
                # the backward kernel's fp16-I/O instance: the graph's to_fp32 / to_fp16 casts
                # (models/FPN/builder.py:581-586, 607-608) happen inside the kernel
                ptrs16, Hs16, Ws16 = self._levels(in_grad[:-1])
                lib().cdll.sd_fpn_roi_align_bwd_workspace_bytes.restype = ctypes.c_size_t
                wsb = lib().cdll.sd_fpn_roi_align_bwd_workspace_bytes(Hs16, Ws16, len(feats), B, rois.shape[1])
                ws = _scratch(rois, wsb)
                try:
                    _call("sd_fpn_roi_align_bwd_packed_f16", _ptr(out_grad[0]), _ptr(rois), _ptr(out_data[1]),
                               _ptr(out_data[2]), ptrs16, Hs16, Ws16, _iarr(self.strides), len(feats), req_data,
                               B, C, rois.shape[1], self.pooled[0], self.pooled[1], float(self.scale0),
                               float(self.lvl0), _ptr(ws), ctypes.c_size_t(wsb), None)
                    _sync()
                    self.assign(in_grad[-1], req[-1], 0)
                    return
                except SimpleDetOpsError as e:
                    if e.code != SD_ERR_UNSUPPORTED:
                        raise
            if self.fp16:
                # the sums are formed by the fp32 kernel: the graph's to_fp32 / to_fp16 casts
                # (models/FPN/builder.py:581-586, 607-608) happen here, at the op boundary
                og = _scratch(rois, out_grad[0].size * 4).reshape(out_grad[0].shape)
                _call("sd_cast_f16_to_f32", _ptr(out_grad[0]), _ptr(og), ctypes.c_size_t(out_grad[0].size), None)
                grads16 = in_grad[:-1]
                in_grad = [_scratch(rois, g.size * 4).reshape(g.shape) for g in grads16] + [in_grad[-1]]
                rq = {REQ["write"]}
            else:
                rq = {req_data}
            ptrs, Hs, Ws = self._levels(in_grad[:-1])
            out_grad = [og]
            if self.packed and not self.fp16 and getattr(self, "_planned", False):
                plan = out_data[3]   # lists / tap tables left by this op's forward
                _call("sd_fpn_roi_align_bwd_packed_plan", _ptr(out_grad[0]), _ptr(rois), _ptr(out_data[1]),
                           _ptr(out_data[2]), ptrs, Hs, Ws, _iarr(self.strides), len(feats), rq.pop(), B, C,
                           rois.shape[1], self.pooled[0], self.pooled[1], float(self.scale0),
                           float(self.lvl0), _ptr(plan), ctypes.c_size_t(plan.size), None)
            elif self.packed:  # with the workspace the per-band RoI lists are built once, not per channel
                lib().cdll.sd_fpn_roi_align_bwd_workspace_bytes.restype = ctypes.c_size_t
                wsb = lib().cdll.sd_fpn_roi_align_bwd_workspace_bytes(Hs, Ws, len(feats), B, rois.shape[1])
                ws = _scratch(rois, wsb)
                _call("sd_fpn_roi_align_bwd_packed_ws", _ptr(out_grad[0]), _ptr(rois), _ptr(out_data[1]),
                           _ptr(out_data[2]), ptrs, Hs, Ws, _iarr(self.strides), len(feats), rq.pop(), B, C,
                           rois.shape[1], self.pooled[0], self.pooled[1], float(self.scale0),
                           float(self.lvl0), _ptr(ws), ctypes.c_size_t(wsb), None)
            else:
                _call("sd_fpn_roi_align_bwd", _ptr(out_grad[0]), _ptr(rois), _ptr(out_data[1]),
                           _ptr(out_data[2]), ptrs, Hs, Ws, _iarr(self.strides), len(feats), rq.pop(), B, C,
                           rois.shape[1], self.pooled[0], self.pooled[1], float(self.scale0),
                           float(self.lvl0), None)
            if self.fp16:
                for g32, g16 in zip(in_grad[:-1], grads16):
                    _call("sd_cast_f32_to_f16", _ptr(g32), _ptr(g16), ctypes.c_size_t(g16.size), req_data, None)
            _sync()
            self.assign(in_grad[-1], req[-1], 0)

    class FPNRoIAlignProp(CustomOpProp):
        def __init__(self, rcnn_stride, pooled_size="(7, 7)", roi_canonical_scale="224",
                     roi_canonical_level="4", fp16="False"):
            super().__init__(need_top_grad=True)
            self.rcnn_stride = _tuple(rcnn_stride, typ=int)
            self.pooled_size = _tuple(pooled_size, 2, int)
            self.scale0, self.lvl0 = float(roi_canonical_scale), float(roi_canonical_level)
            self.packed = _fpn_packed(self.pooled_size)
            self.fp16 = _bool(fp16)
            if self.fp16 and not self.packed:
                raise ValueError("fpn_roi_align: fp16 I/O is provided for 7x7 and 14x14 pooling")

        def list_arguments(self):
            return ["data_s{}".format(s) for s in self.rcnn_stride] + ["rois"]

        def list_outputs(self):
            # packed: output, then private forward -> backward state (arg-max codes, coordinate table,
            # the backward's band lists / tap tables built by the forward's pre-pass)
            return ["output", "argmax", "coords", "plan"] if self.packed else ["output", "maxidx_x", "maxidx_y"]

        num_visible_outputs = 1

        def infer_shape(self, in_shape):
            feats, b = in_shape[:-1], in_shape[-1]
            if len(b) != 3 or b[2] != 4:
                raise ValueError("bbox should be a 3D tensor of shape [batch, rois, 4]")
            o = (b[0], b[1], feats[0][1], self.pooled_size[0], self.pooled_size[1])
            if self.packed:
                stride = int(lib().cdll.sd_fpn_roi_align_argmax_stride(*self.pooled_size))
                lib().cdll.sd_fpn_roi_align_plan_bytes.restype = ctypes.c_size_t
                pb = int(lib().cdll.sd_fpn_roi_align_plan_bytes(_iarr([f[2] for f in feats]),
                                                                _iarr([f[3] for f in feats]), len(feats),
                                                                int(b[0]), int(b[1])))
                return in_shape, [o, (b[0], b[1], feats[0][1], stride),
                                  (b[0], b[1], 9 * (self.pooled_size[0] + self.pooled_size[1])),
                                  ((pb + 15) // 16 * 16,)]
            return in_shape, [o, o, o]

        def infer_type(self, in_type):
            import numpy as np
            f32 = np.float32
            if self.fp16:  # feature maps fp16, rois fp32 -> output fp16; the state keeps its types
                return [np.float16] * (len(in_type) - 1) + [f32], [np.float16, np.uint8, f32, np.uint8], []
            if self.packed:
                return in_type, [f32, np.uint8, f32, np.uint8], []
            return in_type, [f32, f32, f32], []

        def create_operator(self, ctx, shapes, dtypes):
            return FPNRoIAlign(self.rcnn_stride, self.pooled_size, self.scale0, self.lvl0, self.fp16)

        def declare_backward_dependency(self, out_grad, in_data, out_data):
            return [out_grad[0], in_data[-1]] + list(out_data[1:])

    ops["fpn_roi_align"] = (FPNRoIAlignProp, None)

    # ---- _contrib_Proposal_v3: cls_prob, bbox_pred, im_info -> output [, score] ----
    class ProposalV3(CustomOp):
        def __init__(self, g):
            super().__init__()
            self.g = g

        def forward(self, is_train, req, in_data, out_data, aux):
            _no_add(req)
            cls_prob, bbox_pred, im_info = in_data
            _wait(cls_prob, bbox_pred, im_info)
            g = self.g
            B, A2, H, W = cls_prob.shape
            A = A2 // 2
            wsb = lib().cdll.sd_proposal_v3_workspace_bytes(B, A, H, W, g["pre"])
            ws = _scratch(cls_prob, wsb)
            count = A * H * W
            pre = min(g["pre"] if g["pre"] > 0 else count, count)
            if g["is_train"] and g["post"] > pre:
                # same quirk as _contrib_NMS (proposal_v3.cu:474-479,626-632): declared
                # (B, post, .), written min(post, pre) rows per image, packed; tail zeroed here
                self.assign(out_data[0], "write", 0)
                self.assign(out_data[1], "write", 0)
            fa = lambda v: (ctypes.c_float * len(v))(*v)
            _call("sd_proposal_v3_iou" if g["iou_loss"] else "sd_proposal_v3",
                       _ptr(cls_prob), _ptr(bbox_pred), _ptr(im_info),
                       _ptr(out_data[0]), _ptr(out_data[1]), B, A, H, W, g["pre"], g["post"],
                       float(g["thr"]), g["min_size"], fa(g["scales"]), len(g["scales"]),
                       fa(g["ratios"]), len(g["ratios"]), g["stride"], int(g["is_train"]), _ptr(ws),
                       ctypes.c_size_t(wsb), None)
            _sync()

        def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
            for i in range(3):
                self.assign(in_grad[i], req[i], 0)

    class ProposalV3Prop(CustomOpProp):
        def __init__(self, rpn_pre_nms_top_n="6000", rpn_post_nms_top_n="300", threshold="0.7",
                     rpn_min_size="16", scales="(4,8,16,32)", ratios="(0.5,1,2)",
                     feature_stride="16", output_score="False", iou_loss="False", is_train="False",
                     workspace="256"):
            super().__init__(need_top_grad=False)
            self.g = dict(pre=int(rpn_pre_nms_top_n), post=int(rpn_post_nms_top_n),
                          thr=float(threshold), min_size=int(rpn_min_size), scales=_tuple(scales),
                          ratios=_tuple(ratios), stride=int(feature_stride),
                          is_train=_bool(is_train), iou_loss=_bool(iou_loss))  # proposal_v3.cu:536
            self.num_visible_outputs = 2 if _bool(output_score) else 1

        def list_arguments(self):
            return ["cls_prob", "bbox_pred", "im_info"]

        def list_outputs(self):
            return ["output", "score"]

        def infer_shape(self, in_shape):
            d = in_shape[0]
            if len(d) != 4:
                raise ValueError("cls_prob should be (batch, 2 * num_anchors, H, W)")
            g = self.g
            A = d[1] // 2
            if A != len(g["scales"]) * len(g["ratios"]):
                raise ValueError("num_anchors != len(ratios) * len(scales)")
            count = A * d[2] * d[3]
            post = g["post"]  # ProposalProp_v3::InferShape (proposal_v3-inl.h:199-218)
            return [d, (d[0], 4 * A, d[2], d[3]), (d[0], 3)], [(d[0], post, 4), (d[0], post, 1)]

        def create_operator(self, ctx, shapes, dtypes):
            return ProposalV3(self.g)

        def declare_backward_dependency(self, out_grad, in_data, out_data):
            return []

    ops["_contrib_Proposal_v3"] = (ProposalV3Prop, ("contrib", "Proposal_v3"))

    # ---- get_top_proposal (models/FPN/get_top_proposal.py): bbox, score -> top_n of each ----
    class GetTopProposal(CustomOp):
        def __init__(self, top_n):
            super().__init__()
            self.top_n = top_n

        def forward(self, is_train, req, in_data, out_data, aux):
            _no_add(req)
            bbox, score = in_data
            _wait(bbox, score)
            _call("sd_get_top_proposal", _ptr(bbox), _ptr(score), bbox.shape[0], bbox.shape[1],
                       self.top_n, _ptr(out_data[0]), _ptr(out_data[1]), None)
            _sync()

        def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
            self.assign(in_grad[0], req[0], 0)
            self.assign(in_grad[1], req[1], 0)

    class GetTopProposalProp(CustomOpProp):
        def __init__(self, top_n):
            super().__init__(need_top_grad=False)
            self.top_n = int(top_n)

        def list_arguments(self):
            return ["bbox", "score"]

        def list_outputs(self):
            return ["bbox", "score"]

        def infer_shape(self, in_shape):
            b = in_shape[0]
            return in_shape, [(b[0], self.top_n, b[2]), (b[0], self.top_n, 1)]

        def create_operator(self, ctx, shapes, dtypes):
            return GetTopProposal(self.top_n)

        def declare_backward_dependency(self, out_grad, in_data, out_data):
            return []

    ops["get_top_proposal"] = (GetTopProposalProp, None)

    # ---- _contrib_DecodeBBox: rois, bbox_pred, im_info -> output ----
    class DecodeBBox(CustomOp):
        def __init__(self, g):
            super().__init__()
            self.g = g

        def forward(self, is_train, req, in_data, out_data, aux):
            _no_add(req)
            _require_write(req[:1], ["output"])
            rois, pred, info = in_data
            _wait(rois, pred, info)
            g = self.g
            fa = lambda v: (ctypes.c_float * 4)(*v)
            _call("sd_decode_bbox", _ptr(rois), _ptr(pred), _ptr(info), _ptr(out_data[0]),
                       rois.shape[0], rois.shape[1], pred.shape[2] // 4, fa(g["mean"]), fa(g["std"]),
                       int(g["agnostic"]), int(g["xyxy"]), None)
            _sync()

        def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
            for i in range(3):
                self.assign(in_grad[i], req[i], 0)

    class DecodeBBoxProp(CustomOpProp):
        def __init__(self, bbox_mean="(0,0,0,0)", bbox_std="(0.1,0.1,0.2,0.2)",
                     class_agnostic="True", bbox_decode_type="xywh"):
            super().__init__(need_top_grad=False)
            if bbox_decode_type not in ("xywh", "xyxy"):
                raise ValueError("bbox_decode_type must be 'xywh' or 'xyxy'")
            self.g = dict(mean=_tuple(bbox_mean, 4), std=_tuple(bbox_std, 4),
                          agnostic=_bool(class_agnostic), xyxy=bbox_decode_type == "xyxy")

        def list_arguments(self):
            return ["rois", "bbox_pred", "im_info"]

        def list_outputs(self):
            return ["output"]

        def infer_shape(self, in_shape):
            d = in_shape[1]
            out = (d[0], d[1], 4) if self.g["agnostic"] else tuple(d)
            return in_shape, [out]

        def create_operator(self, ctx, shapes, dtypes):
            return DecodeBBox(self.g)

        def declare_backward_dependency(self, out_grad, in_data, out_data):
            return []

    ops["_contrib_DecodeBBox"] = (DecodeBBoxProp, ("contrib", "DecodeBBox"))
    return ops


# ------------------------------------------------------------------------------- registration ----
def register(mx=None):
    """Register every CustomOp (op_type = 'sd_' + reference op name).  Returns {name: PropClass}."""
    if mx is None:
        import mxnet as mx  # noqa: F811  (lazy: MXNet is only needed here)
    lib()  # fail loudly now if the HIP library is missing
    _state["mx"] = mx
    table = _build_ops(mx)
    out = {}
    for name, (prop, _) in table.items():
        out[name] = mx.operator.register(_PREFIX + name)(prop)
    _state["registered"] = True
    _state["table"] = table
    return out


def _namespaces(mx, ns):
    """Every namespace object the reference reaches an operator through: `mx.sym` (= `mx.symbol`) or
    `mx.sym.contrib`, and for contrib operators also the old `mx.contrib.symbol` / `mx.contrib.sym`
    module, which MXNet 1.x still fills with the same constructors as separate attributes
    (models/tridentnet/resnet_v1.py:85 builds its DeformableConvolution through it)."""
    out = []

    def add(t):
        if t is not None and all(t is not o for o in out):
            out.append(t)
    for root in (getattr(mx, "sym", None), getattr(mx, "symbol", None)):
        if root is not None:
            add(getattr(root, ns) if ns else root)
    if ns == "contrib":
        old = getattr(mx, "contrib", None)
        for a in ("symbol", "sym"):
            add(getattr(old, a, None) if old is not None else None)
    return out


def install(mx=None, stream=None, sync=True):
    """register() + alias the reference's symbol constructors to mx.sym.Custom, e.g.
    mx.sym.contrib.ROIAlign_v2(data=d, rois=r, pooled_size=(7,7), spatial_scale=0.25) builds
    mx.sym.Custom(d, r, op_type='sd__contrib_ROIAlign_v2', pooled_size='(7, 7)', ...) and returns
    only the visible outputs, so symbol/builder.py and the config/ graphs stay unchanged.

    The constructor an alias replaces is kept (`<namespace>._sd_reference_<name>`, and on the alias as
    `_sd_original`).  An operator whose prop class has `sd_supports(params)` (DeformableConvolution) hands
    a call with parameters the kernels do not take BACK to that constructor: the node is then the native
    operator, exactly what the graph held without install() (`_state["fallbacks"]` lists them).

    `stream` / `sync`: the hipStream_t every operator launches on (None: the NULL stream; an int; or a callable
    evaluated per call) and whether forward() / backward() synchronise it before they return (the module
    docstring has the ordering contract; sync=False is for a host that orders the outputs on `stream` itself)."""
    props = register(mx)
    mx = _state["mx"]
    _state["fallbacks"] = []
    _state["stream"], _state["sync"] = stream, bool(sync)

    def make(name, prop, original):
        def ctor(*args, **kwargs):
            # MXNet's generated constructors skip inputs / attributes given as None
            # (models/sepc/sepc_dconv.py:13: `bias=bias if not no_bias else None`)
            kwargs = {k: v for k, v in kwargs.items() if v is not None}
            name_kw = kwargs.pop("name", None)
            params = {k: _param_str(v) for k, v in kwargs.items() if not _is_symbol(mx, v)}
            inputs = {k: v for k, v in kwargs.items() if _is_symbol(mx, v)}
            supports = getattr(prop, "sd_supports", None)
            why = supports(params) if supports is not None else ""
            if why:
                if original is None:
                    raise ValueError("%s: %s (and no native constructor to fall back to)" % (name, why))
                _state["fallbacks"].append((name, name_kw, why))
                native = {k: v for k, v in kwargs.items() if k != "cache_col"}
                if name_kw is not None:
                    native["name"] = name_kw
                return original(*args, **native)
            p = prop(**params)
            if args and inputs:
                # MXNet composes a variadic operator (Custom is `*data`) from positional OR keyword Symbols, never
                # both (Symbol._compose raises TypeError) -- and the reference mixes them for the built-in it
                # believes it is calling (models/sepc/sepc_dconv.py:12-16: DeformableConvolution(x, offset,
                # weight=weight, bias=bias, ...)): positional inputs take the operator's argument names in order
                names = list(p.list_arguments())
                if len(args) > len(names):
                    raise TypeError("%s takes %d inputs (%s), %d given positionally" % (name, len(names), names, len(args)))
                for k, v in zip(names, args):
                    if k in inputs:
                        raise TypeError("%s: input '%s' given positionally and by keyword" % (name, k))
                    inputs[k] = v
                # keyword composition matches by name; keep the operator's own order for readability of the graph
                inputs = {k: inputs[k] for k in names if k in inputs} | {k: v for k, v in inputs.items() if k not in names}
                args = ()
            sym = mx.sym.Custom(*args, op_type=_PREFIX + name, name=name_kw, **inputs, **params)
            nvis = getattr(p, "num_visible_outputs", len(p.list_outputs()))
            nout = len(p.list_outputs())
            if nvis == nout:
                return sym
            return sym[0] if nvis == 1 else mx.sym.Group([sym[i] for i in range(nvis)])
        ctor.__name__ = name
        ctor._sd_alias = True
        ctor._sd_original = original
        return ctor

    for name, (prop, where) in _state["table"].items():
        if where is None:
            continue
        ns, attr = where
        for target in _namespaces(mx, ns):
            cur = getattr(target, attr, None)
            # a second install(): the first one's original stays the original
            original = cur._sd_original if getattr(cur, "_sd_alias", False) else cur
            setattr(target, attr, make(name, props[name], original))
            setattr(target, "_sd_reference_" + attr, original)
    # the fused FPN extractor has no single reference symbol to alias: rebind the builder method
    # that emits the subgraph (no reference file is edited)
    _state["fpn_patched"] = patch_fpn_roi_align(mx=mx)
    # ... and the mxnext wrappers the reference's builders go through, explicitly (not relying on
    # mxnext looking `mx.sym.*` up at call time)
    _state["mxnext_patched"] = patch_mxnext(mx=mx)
    return props


def _head_op(mx, sym):
    """(operator name, attrs) of the node behind a symbol: real MXNet through the graph JSON (a
    multi-output symbol's heads all point at one node here), the graph-recording test stub through
    `op_type`.  CustomOps report ('Custom', {'op_type': ...})."""
    if hasattr(sym, "tojson"):
        try:
            import json
            g = json.loads(sym.tojson())
            node = g["nodes"][g["heads"][0][0]]
            return node.get("op"), dict(node.get("attrs", node.get("attr", node.get("param", {}))) or {})
        except Exception:
            pass
    node = sym
    while getattr(node, "op_type", None) in ("_output", "Group"):
        node = node.parent if node.op_type == "_output" else node.inputs[0]
    return getattr(node, "op_type", None), dict(getattr(node, "params", {}) or {})


def patch_mxnext(mxnext=None, mx=None):
    """Point the mxnext wrappers on the hot path at the aliased symbol constructors WHERE THAT IS KNOWN TO
    BE A NO-OP FOR THE GRAPH'S MEANING.  mxnext (github.com/RogerChern/mxnext) is not part of the
    reference tree, so which operator a wrapper builds is not assumed: each saved original is called
    once with placeholder Variables and the reference's own keyword arguments, and the operator of the
    node it returns decides --
      * already an `sd_*` Custom node (the wrapper looks `mx.sym.*` up at call time): nothing to do;
      * the native operator this plugin replaces under the same name (`_contrib_ROIAlign_v2`,
        `ProposalTarget`, `_contrib_Proposal_v3`, `_contrib_DecodeBBox`): the wrapper captured the
        constructor before install(); it is rebound to a function with the call sites' signature that
        builds the alias;
      * anything else (`_contrib_Proposal`, `_contrib_Proposal_v2`, `MultiProposal`, a TVM op ...), or a
        probe that raises: LEFT ALONE -- `proposal.cu`, `proposal_v2.cu` and `proposal_v3.cu` differ in the
        +1 box convention, the dw / dh clamp and the min-size filter, rebinding would change the RPN's
        numbers silently.
    Call sites whose signatures the probes use:
        X.roi_align(feat, rois=, out_size=, stride=, name=)      symbol/builder.py:885, models/FPN/builder.py:592
        X.proposal_target(rois=, gt_boxes=, ..., name=)           symbol/builder.py:304, models/FPN/builder.py:347
        X.proposal(cls_prob=, bbox_pred=, im_info=, ..., iou_loss=, output_score=)   symbol/builder.py:241
        X.decode_bbox(rois=, bbox_pred=, im_info=, ..., name=)    symbol/builder.py:384
        mxnext.tvm.get_top_proposal.get_top_proposal(F, bbox=, score=, top_n=, batch_size=)
                                                                  models/FPN/builder.py:319-321
    (mxnext.tvm.proposal -- the "nnvm" proposal some configs select for the fine levels -- is left
    alone: it is un-vendored and nothing in the reference tree pins its results.)
    Returns the list of rebound names ([] when mxnext is not importable); `_state["mxnext_probe"]`
    holds what every probe saw."""
    mx = mx or _state["mx"]
    if mxnext is None:
        try:
            import importlib
            mxnext = importlib.import_module("mxnext")
        except Exception:
            return []
    done = []
    seen = _state["mxnext_probe"] = {}
    V = lambda n: mx.sym.Variable("_sd_probe_" + n)

    def roi_align(feat, rois, out_size, stride, name=None, **kw):
        return mx.sym.contrib.ROIAlign_v2(data=feat, rois=rois, pooled_size=(int(out_size), int(out_size)),
                                          spatial_scale=1.0 / stride, name=name, **kw)

    def proposal_target(**kw):
        return mx.sym.ProposalTarget(**kw)

    def proposal(**kw):
        return mx.sym.contrib.Proposal_v3(**kw)

    def decode_bbox(**kw):
        return mx.sym.contrib.DecodeBBox(**kw)

    # wrapper -> (replacement, the native operator it must turn out to build, a probe call)
    table = {
        "roi_align": (roi_align, "_contrib_ROIAlign_v2",
                      lambda f: f(V("feat"), rois=V("rois"), out_size=7, stride=16, name="_sd_probe")),
        "proposal_target": (proposal_target, "ProposalTarget",
                            lambda f: f(rois=V("rois"), gt_boxes=V("gt"), num_classes=81, class_agnostic=False,
                                        batch_images=1, proposal_without_gt=False, image_rois=64, fg_fraction=0.25,
                                        fg_thresh=0.5, bg_thresh_hi=0.5, bg_thresh_lo=0.0, bbox_weight=(1., 1., 1., 1.),
                                        bbox_mean=(0., 0., 0., 0.), bbox_std=(.1, .1, .2, .2), name="_sd_probe")),
        "proposal": (proposal, "_contrib_Proposal_v3",
                     lambda f: f(cls_prob=V("cls"), bbox_pred=V("box"), im_info=V("info"), name="_sd_probe",
                                 feature_stride=16, scales=(8,), ratios=(0.5, 1.0, 2.0), rpn_pre_nms_top_n=12,
                                 rpn_post_nms_top_n=6, threshold=0.7, rpn_min_size=0, iou_loss=False,
                                 output_score=True)),
        "decode_bbox": (decode_bbox, "_contrib_DecodeBBox",
                        lambda f: f(rois=V("rois"), bbox_pred=V("box"), im_info=V("info"), name="_sd_probe",
                                    bbox_mean=(0., 0., 0., 0.), bbox_std=(.1, .1, .2, .2), class_agnostic=False)),
    }
    for attr, (fn, native, probe) in table.items():
        cur = getattr(mxnext, attr, None)
        if cur is None:
            continue
        orig = cur._sd_original if getattr(cur, "_sd_alias", False) else cur   # (a second install())
        try:
            op, attrs = _head_op(mx, probe(orig))
        except Exception as e:
            seen[attr] = "probe failed: %s" % (e,)
            continue
        short = native[len("_contrib_"):] if native.startswith("_contrib_") else native
        if op == "Custom" and str(attrs.get("op_type", "")).startswith(_PREFIX) or str(op).startswith(_PREFIX):
            seen[attr] = "late binding: already builds %s" % (attrs.get("op_type", op),)
            continue
        if op not in (native, short):
            seen[attr] = "builds %s, not %s: left alone" % (op, native)
            continue
        seen[attr] = "builds %s at import-time binding: rebound" % (op,)
        fn._sd_alias, fn._sd_original = True, orig
        try:
            setattr(mxnext, "_sd_reference_" + attr, orig)
            setattr(mxnext, attr, fn)
            done.append("mxnext." + attr)
        except Exception:
            pass
    try:
        import importlib
        m = importlib.import_module("mxnext.tvm.get_top_proposal")
        cur = m.get_top_proposal
        orig = cur._sd_original if getattr(cur, "_sd_alias", False) else cur

        def get_top_proposal(F, bbox, score, top_n, batch_size=None, name="get_top_proposal", **kw):
            # TWO results, (bbox, score), as every call site unpacks them: models/FPN/builder.py:319-323 returns
            # the wrapper's result from get_all_proposal(), and :345 (and symbol/builder.py:36,92,302,
            # models/maskrcnn/builder.py:113,182, ...: all 18 callers) does
            # `(proposal, proposal_score) = self.get_all_proposal(...)` before `rois=proposal`.  A plain tuple
            # of two single-output symbols: nothing multi-output can reach an operator argument by accident.
            sym = mx.sym.Custom(bbox=bbox, score=score, op_type=_PREFIX + "get_top_proposal",
                                top_n=_param_str(top_n), name=name)
            return sym[0], sym[1]
        get_top_proposal._sd_alias, get_top_proposal._sd_original = True, orig
        m._sd_reference_get_top_proposal = orig
        m.get_top_proposal = get_top_proposal
        done.append("mxnext.tvm.get_top_proposal.get_top_proposal")
    except Exception:
        pass
    return done


def patch_fpn_roi_align(builder_module=None, mx=None):
    """Route the reference's FPN RoI extractor to the fused op WITHOUT editing the reference:
    rebinds `models.FPN.builder.FPNRoiAlign.get_roi_feature` (models/FPN/builder.py:567-610: fpn_roi_assign
    -> one X.roi_align per stride -> reshape -> add_n) to a method that emits ONE
    mx.sym.Custom(op_type='sd_fpn_roi_align') node over the same inputs and returns the same
    (B*R, C, out, out) symbol (fp16 graphs: cast to fp32 before, back to fp16 after, as :581-586,
    607-608).  install() calls this when `models.FPN.builder` is importable; returns True when the
    class was patched.  The original method is kept as `_sd_reference_get_roi_feature`."""
    mx = mx or _state["mx"]
    if builder_module is None:
        try:
            import importlib
            builder_module = importlib.import_module("models.FPN.builder")
        except Exception:
            return False
    cls = getattr(builder_module, "FPNRoiAlign", None)
    if cls is None or getattr(cls, "_sd_patched", False):
        return cls is not None

    def get_roi_feature(self, conv_fpn_feat, proposal):
        p = self.p
        strides = tuple(int(s) for s in p.stride)
        out = int(p.out_size)
        fp16 = bool(getattr(p, "fp16", False))
        native16 = fp16 and out in (7, 14)   # the op reads / writes fp16 itself: no cast nodes
        feats = []
        for s_ in strides:
            f = conv_fpn_feat["stride%s" % s_]
            if fp16 and not native16:
                f = mx.sym.Cast(data=f, dtype="float32", name="fpn_stride%s_to_fp32" % s_)
            feats.append(f)
        extra = {"fp16": "True"} if native16 else {}
        sym = mx.sym.Custom(*feats, proposal, op_type=_PREFIX + "fpn_roi_align",
                            rcnn_stride=_param_str(strides), pooled_size=_param_str((out, out)),
                            roi_canonical_scale=_param_str(p.roi_canonical_scale),
                            roi_canonical_level=_param_str(p.roi_canonical_level), name="fpn_roi_align",
                            **extra)
        roi_feat = mx.sym.reshape(data=sym[0], shape=(-3, -2), name="roi_feat_reshape")
        if fp16 and not native16:
            roi_feat = mx.sym.Cast(data=roi_feat, dtype="float16", name="roi_feat_to_fp16")
        return roi_feat

    cls._sd_reference_get_roi_feature = cls.get_roi_feature
    cls.get_roi_feature = get_roi_feature
    cls._sd_patched = True
    return True


def _param_str(v):
    """keyword argument -> the string MXNet's front end would send (str(value)); numpy scalars
    are unwrapped first (numpy 2 prints np.float32(0.25) for repr)."""
    if isinstance(v, str):
        return v
    if isinstance(v, (tuple, list)):
        return "(" + ", ".join(_param_str(x) for x in v) + ("," if len(v) == 1 else "") + ")"
    if hasattr(v, "item") and not isinstance(v, (bool, int, float)):
        v = v.item()
    return repr(v) if isinstance(v, float) else str(v)


def _is_symbol(mx, v):
    sym_t = getattr(getattr(mx, "sym", None), "Symbol", None)
    return sym_t is not None and isinstance(v, sym_t)
