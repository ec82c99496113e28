/*
 * simpledet_ops.h -- C ABI of libsimpledet_ops_hip.so: MI355X (gfx950) kernels for the
 * second-stage detection-ops hot path of tusen-ai/simpledet.
 *
 * This is the drop-in boundary.  The reference has no C ABI of its own: its operators are compiled
 * into libmxnet and reached by *operator name* (NNVM FCompute / legacy OperatorProperty), and its
 * only run-time extension hook is the Python mx.operator.CustomOp.  Each entry point below is what
 * a CustomOp (simpledet_amd/mxnet_plugin.py) or an FCompute<gpu> shim binds for one reference
 * operator; the comment above each names the reference interface it replaces (file:line relative to
 * the simpledet tree).
 *
 * Conventions
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer unless the name ends in _host
 *   - the caller owns every buffer (incl. workspaces: ask sd_*_workspace_bytes first); the library
 *     allocates nothing on the device and keeps no state between calls
 *   - `stream` is a hipStream_t (NULL = default stream); calls are asynchronous on it; no entry
 *     point synchronises the device
 *   - all tensors are dense row-major ("C contiguous"), fp32 unless stated
 *   - `req` mirrors MXNet OpReqType: 0 = kNullOp, 1 = kWriteTo, 3 = kAddTo (kWriteInplace=2 is
 *     rejected exactly as the reference rejects it, roi_align_v2.cu:105-108)
 *   - return 0 on success, negative SD_ERR_* otherwise; sd_last_error() gives the message
 *     (thread local).  Nothing aborts or throws across this boundary.
 */
#ifndef SIMPLEDET_OPS_H_
#define SIMPLEDET_OPS_H_
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SD_OK 0
#define SD_ERR_INVALID_ARG (-1)
#define SD_ERR_UNSUPPORTED (-2)
#define SD_ERR_HIP (-3)
#define SD_ERR_WORKSPACE (-4)

#define SD_REQ_NULL 0
#define SD_REQ_WRITE 1
#define SD_REQ_ADD 3

#define SD_MAX_FPN_LEVELS 8

const char* sd_last_error(void);
/* ABI version of this header (bumped on any signature change) */
int sd_abi_version(void);
/* kernel-variant knobs for A/B measurements (bench.py); unknown keys are an error */
int sd_set_tuning(const char* key, int value);
int sd_get_tuning(const char* key, int* value);

/* HBM streaming copy (measurement aid, no reference counterpart): dst[i] = src[i] with
 * width_bytes (4, 8 or 16) per lane.  bench.py uses it to report the achievable HBM rate next to
 * the 8 TB/s spec peak and to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on a known byte count */
int sd_hbm_stream_copy(const void* src, void* dst, size_t bytes, int width_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ROIAlign_v2  (mx.sym.contrib.ROIAlign_v2, registered as _contrib_ROIAlign_v2)
 *   replaces ROIAlignForward_v2<gpu>  operator_cxx/contrib/roi_align_v2-inl.h:157-195
 *            (kernel ROIAlignForwardKernel_v2::Map :61-153; shapes roi_align_v2.cc:187-208)
 *   data (B,C,H,W)  rois (B,R,4) [x1,y1,x2,y2] image coords, batch index = roi / R
 *   out, maxidx_x, maxidx_y (B,R,C,ph,pw)
 * ---------------------------------------------------------------------------------------------- */
int sd_roi_align_v2_fwd(const float* data, const float* rois, float* out, float* maxidx_x,
                        float* maxidx_y, int B, int C, int H, int W, int R, int pooled_h,
                        int pooled_w, float spatial_scale, void* stream);

/*   replaces ROIAlignBackward_v2<gpu>  operator_cxx/contrib/roi_align_v2.cu:87-143
 *            (kernel ROIAlignBackwardKernelGPU_v2::Map :35-84; inputs per ROIAlignGrad_v2
 *            roi_align_v2-inl.h:206-218: [dY, rois, maxidx_x, maxidx_y] -> [dX, d_rois])
 *   d_data (B,C,H,W) honours req_data (write = zero first, add = accumulate);
 *   d_rois (B,R,4) is zero-filled when req_rois == write (may be NULL when req_rois == null). */
int sd_roi_align_v2_bwd(const float* out_grad, const float* rois, const float* maxidx_x,
                        const float* maxidx_y, float* d_data, float* d_rois, int req_data,
                        int req_rois, int B, int C, int H, int W, int R, int pooled_h, int pooled_w,
                        float spatial_scale, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Fused FPN RoI feature extraction = FPNRoiAlign.get_roi_feature, models/FPN/builder.py:567-610:
 *   fpn_roi_assign (models/FPN/assign_layer_fpn.py:17-41) -> one ROIAlign_v2 per level on the
 *   zero-masked RoIs -> add_n.  One launch reads every level once and writes ONE output
 *   (the reference writes 4 x 3 full-size tensors and adds them).
 *   feats_host: host array of nlvl device pointers, level l is (B,C,Hs[l],Ws[l]), spatial_scale
 *   1/strides[l].  maxidx_x/y hold the argmax of each RoI's assigned level (-1 elsewhere).
 * ---------------------------------------------------------------------------------------------- */
int sd_fpn_roi_align_fwd(const float* const* feats_host, const int* Hs_host, const int* Ws_host,
                         const int* strides_host, int nlvl, const float* rois, float* out,
                         float* maxidx_x, float* maxidx_y, int B, int C, int R, int pooled_h,
                         int pooled_w, float roi_canonical_scale, float roi_canonical_level,
                         void* stream);
int sd_fpn_roi_align_bwd(const float* out_grad, const float* rois, const float* maxidx_x,
                         const float* maxidx_y, float* const* d_feats_host, const int* Hs_host,
                         const int* Ws_host, const int* strides_host, int nlvl, int req_data, int B,
                         int C, int R, int pooled_h, int pooled_w, float roi_canonical_scale,
                         float roi_canonical_level, void* stream);
/* assign_layer_fpn CustomOp (models/FPN/assign_layer_fpn.py:10-73): rois (n_rois,4) ->
 * rois_per_level (nlvl, n_rois, 4) zero-masked, and optionally level (n_rois) int32 (-1 = none) */
int sd_fpn_roi_assign(const float* rois, int n_rois, const int* strides_host, int nlvl,
                      float roi_canonical_scale, float roi_canonical_level, float* rois_per_level,
                      int32_t* level, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SIMPLEDET_OPS_H_ */
