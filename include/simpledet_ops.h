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
 *     allocates nothing on the device and keeps no state between calls (the only process-wide
 *     state is the kernel-variant selection of sd_set_tuning, read lock-free at launch)
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
/* the kernels the calling thread's last fused-RoIAlign entry point launched, e.g.
 * "sd::roi_prep_merged_kernel<7> + sd::roi_align_fwd_band<7,true,false>" (measurement tools label
 * their numbers with the dispatch actually taken instead of assuming one) */
const char* sd_last_dispatch(void);
/* ABI version of this header: bumped on any signature change AND on any change of a buffer layout
 * or size contract (a caller built against an older header must not pass it buffers of the old
 * size).  History: 1 round 1; 2 packed arg-max rows padded to whole dwords (7x7: 49 -> 52 bytes)
 * and 4-byte aligned; 3 sd_fpn_roi_align_workspace_bytes grew (the forward's band lists / tap
 * entries live in the workspace; with a smaller or NULL workspace the forward still runs, on the
 * slower tiled kernels); 4 sd_gemm_f32 computes products as three bf16 MFMA terms by default
 * (same signature, documented error model), sd_proposal_mask_target_ratio / sd_cast_* / *_f16 added;
 * 5 sd_gemm_f32_ws, the DCN products default to the scaled fp16 split (fp32-path accuracy), plain
 * sd_gemm_f32 to exact fp32.
 * sd_abi_version() returns the library's value; compare with this macro. */
#define SD_ABI_VERSION 7
int sd_abi_version(void);
/* kernel-variant knobs for A/B measurements (bench.py, tests); every variant computes the same
 * result.  Unknown keys are an error.  Knobs that disable parts of a kernel for profiling exist
 * only in the separate -DSD_PROFILING build used by tools/, not in this library. */
int sd_set_tuning(const char* key, int value);
int sd_get_tuning(const char* key, int* value);

/* Block the calling host thread until `stream` has drained.  Only the MXNet CustomOp adapter
 * uses it (a CustomOp's outputs must be complete when forward()/backward() returns); the compute
 * entry points never synchronise. */
int sd_stream_synchronize(void* stream);

/* HBM streaming copy (measurement aid, no reference counterpart): dst[i] = src[i] with
 * width_bytes (4, 8 or 16) per lane.  bench.py uses it to report the achievable HBM rate next to
 * the 8 TB/s spec peak and to calibrate rocprofv3's FETCH_SIZE / WRITE_SIZE on a known byte count */
int sd_hbm_stream_copy(const void* src, void* dst, size_t bytes, int width_bytes, void* stream);

/* fp16 <-> fp32 streaming casts (X.to_fp32 / X.to_fp16 of the reference's fp16 graphs, where an op
 * boundary still needs them: the fp16 form of the RoIAlign backward).  n elements; dst of the
 * second honours req (write / add, the sum formed in fp32).  16-byte aligned buffers for the first. */
int sd_cast_f16_to_f32(const void* src, float* dst, size_t n, void* stream);
int sd_cast_f32_to_f16(const float* src, void* dst, size_t n, int req, void* stream);

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

/* The same with DEVICE scratch of sd_roi_align_v2_workspace_bytes(B, R) bytes: the forward then runs
 * on the band-resident kernel (feature planes streamed through LDS once, no per-RoI gathers; same
 * bits).  workspace may be NULL (= the call above, the tiled kernels). */
size_t sd_roi_align_v2_workspace_bytes(int B, int R);
int sd_roi_align_v2_fwd_ws(const float* data, const float* rois, float* out, float* maxidx_x,
                           float* maxidx_y, int B, int C, int H, int W, int R, int pooled_h,
                           int pooled_w, float spatial_scale, void* workspace, size_t workspace_bytes,
                           void* stream);

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
                         void* workspace, size_t workspace_bytes, void* stream);
/* DEVICE scratch for the forward: the band-resident kernel (the default: planes streamed through
 * LDS once, no per-RoI gathers) keeps its per-band item lists and per-RoI tap entries there; with
 * workspace == NULL (or too small) the forward runs on the tiled kernels instead -- same bits */
size_t sd_fpn_roi_align_workspace_bytes(int B, int R);
int sd_fpn_roi_align_bwd(const float* out_grad, const float* rois, const float* maxidx_x,
                         const float* maxidx_y, float* const* d_feats_host, const int* Hs_host,
                         const int* Ws_host, const int* strides_host, int nlvl, int req_data, int B,
                         int C, int R, int pooled_h, int pooled_w, float roi_canonical_scale,
                         float roi_canonical_level, void* stream);
/* The same fused op with the arg-max kept as ONE byte per output (row sample * 3 + column sample,
 * 255 = nothing pooled) plus a small per-RoI table (coords: B*R x 9*(pooled_h+pooled_w) 4-byte
 * words, 516 KB at the baseline; per RoI: 3*(ph+pw) fp32 sample coordinates, then 3*(ph+pw) pairs
 * {clamped neighbours lo | hi << 16, interpolation fraction} derived from them with the backward's
 * own expressions) instead of two fp32 planes: the arg-max is state between this op's own forward
 * and backward, not part of the reference's graph interface.  The backward looks the pair up, i.e.
 * uses exactly the values the float planes would have produced, without floor / clamp / divide
 * per gradient element.  Cuts the forward's writes from 3 to 1.25 planes and the backward's reads
 * likewise (7x7 and 14x14 pooling).  coords must be 8-byte aligned.
 * argmax layout: (B, R, C, sd_fpn_roi_align_argmax_stride(ph, pw)) bytes -- every (RoI, channel)
 * row of ph*pw codes is padded to whole 4-byte words (7x7: 52, 14x14: 196) so that the backward
 * fetches four codes with one aligned load; the padding bytes are never read as codes.  argmax
 * must be 4-byte aligned. */
int sd_fpn_roi_align_argmax_stride(int pooled_h, int pooled_w);
int sd_fpn_roi_align_fwd_packed(const float* const* feats_host, const int* Hs_host,
                                const int* Ws_host, const int* strides_host, int nlvl,
                                const float* rois, float* out, uint8_t* argmax, float* coords, int B,
                                int C, int R, int pooled_h, int pooled_w, float roi_canonical_scale,
                                float roi_canonical_level, void* workspace, size_t workspace_bytes,
                                void* stream);
/* fp16 I/O variant for fp16 graphs (the reference casts to fp32 around the op, models/FPN/builder.py:
 * 581-586, 607-608): feats are fp16 (B,C,H,W), out is fp16 (B,R,C,ph,pw); the arithmetic is the fp32
 * arithmetic of the op above (taps converted exactly, the maximum rounded to nearest even), i.e.
 * bit-equal to to_fp32 -> sd_fpn_roi_align_fwd_packed -> to_fp16 without the two cast passes and with
 * half the feature traffic.  argmax / coords / workspace as above; the workspace is REQUIRED (the fp16
 * path exists on the band-resident kernel only), feats must be 16-byte aligned. */
int sd_fpn_roi_align_fwd_packed_f16(const void* const* feats_host, const int* Hs_host,
                                    const int* Ws_host, const int* strides_host, int nlvl,
                                    const float* rois, void* out, uint8_t* argmax, float* coords, int B,
                                    int C, int R, int pooled_h, int pooled_w, float roi_canonical_scale,
                                    float roi_canonical_level, void* workspace, size_t workspace_bytes,
                                    void* stream);
int sd_fpn_roi_align_bwd_packed(const float* out_grad, const float* rois, const uint8_t* argmax,
                                const float* coords, float* const* d_feats_host, const int* Hs_host, const int* Ws_host,
                                const int* strides_host, int nlvl, int req_data, int B, int C, int R,
                                int pooled_h, int pooled_w, float roi_canonical_scale,
                                float roi_canonical_level, void* stream);
/* Numerics of the packed backward (and of the drop-in sd_roi_align_v2_bwd / sd_fpn_roi_align_bwd in
 * their default form).  Every tap value is computed in fp32 exactly as the reference does; the SUM a
 * pixel receives is accumulated in 32-bit fixed point whose unit is fixed per workgroup (one band of
 * one channel): 2^-30 .. 2^-29 of (max|dY| of the band) x (a bound on the weight one pixel can collect).  With
 * a workspace the bound is per PIXEL of the band (round 6: a 2-D difference array of the RoIs' footprints in the
 * list pre-pass, typically 20-100), without one it is summed over the band's RoIs (1,000-1,800 at the baseline).
 * The result is independent of the order of the adds (bit-reproducible; the reference's float atomics are not).
 *   Measured against the oracle (tests/test_fixed_point_precision.py, profiles/r06e_fixed_point_precision.json):
 * dY ~ N(0,1) at BASELINE's size: 1.1e-5 max abs error with the per-pixel bound (2.3e-5 with the summed bound,
 * 3.8e-6 for fp32 adds in hardware order; bar 1e-4).
 *   A fixed-point unit is ABSOLUTE for its band, so it is only used while the band's gradients span a range
 * it resolves (round 6): behind the scatter a workgroup compares the exponent of its largest |dY| plus the bits
 * of its weight bound with the MEAN exponent of its non-zero gradients (a quarter of them, sampled at fixed
 * positions; all-integer, so the verdict does not depend on any order) and keeps the integer sums only if
 *     E(max|dY|) + ceil(log2(weight bound)) <= mean E(dY) + 15,
 * i.e. the unit is below ~2^-13 of the gradients' geometric mean.  Near-Gaussian gradients pass with 4-9 bits to
 * spare (a loss scale cancels out); heavy-tailed ones -- dY = N(0,1) x lognormal(sigma = 3) x 128: until round 6
 * 39 % of the elements were off by more than 1e-4 relative, the median small element by 0.8 % -- make the
 * workgroup clear its band and sum it again with fp32 compare-and-swap adds, the reference's own arithmetic
 * (roi_align_v2.cu:67-83): then 9e-7 of the elements differ from the oracle by more than 1e-4 max(1, |want|), the
 * same as for the float adds alone (two float summation orders differ where a pixel's addends cancel), and
 * |err| <= 1e-4 max(median|dY|, sum of the pixel's |addends|) holds everywhere.  Non-finite dY (inf AND NaN:
 * the maximum is taken over bit patterns) and weight bounds above 2048 take the float adds as well.  Callers
 * that want fp32 adds in every workgroup select them with
 *     sd_set_tuning("roi_align_bwd_fx", 0)
 * (packed arg-max and float arg-max planes alike) at ~1.2-1.5 x the time; the sums then depend on the order
 * in which the hardware serves the adds, like the reference's.
 *   sd_roi_align_v2_bwd on a single map with C % 4 == 0 whose four planes fit 72 KB of LDS (the C4
 * family; square 7x7 / 14x14 pools) runs roi_align_bwd_flt4_kernel (four whole planes per workgroup), which sums
 * the same way, scaled by max|dY| of the workgroup x a per-pixel weight bound taken from
 * the RoIs' footprints (so `rois` must be the boxes the arg-max planes were produced with, as they are
 * in the operator): bit-reproducible, 4.9e-5 from the exact sums for dY ~ N(0,1) at the full C4 shape
 * (2,1024,50,84) x 512 RoIs; the same verdict, the same fall-backs and the same
 * sd_set_tuning("roi_align_bwd_fx", 0) as above.  sd_set_tuning("roi_align_bwd_flt4", 0) selects the banded
 * kernel there. */
/* The same with a device workspace of sd_fpn_roi_align_bwd_workspace_bytes(): the per-band RoI
 * lists are then built by one small pre-pass instead of by every channel's workgroup (same
 * results bit for bit).  workspace may be NULL (= the call above). */
size_t sd_fpn_roi_align_bwd_workspace_bytes(const int* Hs_host, const int* Ws_host, int nlvl, int B, int R);
int sd_fpn_roi_align_bwd_packed_ws(const float* out_grad, const float* rois, const uint8_t* argmax,
                                   const float* coords, float* const* d_feats_host, const int* Hs_host,
                                   const int* Ws_host, const int* strides_host, int nlvl, int req_data, int B,
                                   int C, int R, int pooled_h, int pooled_w, float roi_canonical_scale,
                                   float roi_canonical_level, void* workspace, size_t workspace_bytes,
                                   void* stream);
/* fp16 gradient in, fp16 feature gradients out (the backward of sd_fpn_roi_align_fwd_packed_f16): the
 * sums are formed exactly as in the fp32 call -- fp32 tap values, fixed-point / fp32 accumulation --
 * only the two conversions of the graph's to_fp32 / to_fp16 casts (models/FPN/builder.py:581-586,
 * 607-608) happen inside the kernel: bit-equal to cast -> sd_fpn_roi_align_bwd_packed_ws -> cast with
 * req (write / add, the sum formed in fp32).  d_feats 8-byte aligned.  SD_ERR_UNSUPPORTED where the
 * default wide kernel does not apply (callers then use the casts). */
int sd_fpn_roi_align_bwd_packed_f16(const void* out_grad, const float* rois, const uint8_t* argmax,
                                    const float* coords, void* const* d_feats_host, const int* Hs_host,
                                    const int* Ws_host, const int* strides_host, int nlvl, int req_data,
                                    int B, int C, int R, int pooled_h, int pooled_w,
                                    float roi_canonical_scale, float roi_canonical_level, void* workspace,
                                    size_t workspace_bytes, void* stream);
/* ONE rois-only pre-pass per training step.  Both pre-passes of the fused extractor -- the forward's
 * item lists / tap entries / coordinate table and the backward's band lists / tap tables -- are pure
 * functions of `rois` and the level geometry.  With a `plan` buffer of sd_fpn_roi_align_plan_bytes()
 * (16-byte aligned, op state between forward and backward like argmax / coords; ~9 MB at the
 * baseline) the forward builds both in its single pre-pass launch and the backward launches its main
 * kernel only: 3 launches per step instead of 4.  Same bits as the _ws pair.  The plan is valid for the
 * shapes, rois and tuning knobs of the forward that filled it; where the band / tap-table form does
 * not apply (knob roi_align_bwd_lists != 1, a level that does not fit) both calls fall back to the
 * behaviour of the unplanned pair by the same deterministic decision.
 *   replaces the same reference code as sd_fpn_roi_align_fwd_packed / _bwd_packed
 *   (models/FPN/builder.py:567-610; roi_align_v2-inl.h:61-195, roi_align_v2.cu:35-143) */
size_t sd_fpn_roi_align_plan_bytes(const int* Hs_host, const int* Ws_host, int nlvl, int B, int R);
int sd_fpn_roi_align_fwd_packed_plan(const float* const* feats_host, const int* Hs_host, const int* Ws_host,
                                     const int* strides_host, int nlvl, const float* rois, float* out,
                                     uint8_t* argmax, float* coords, int B, int C, int R, int pooled_h,
                                     int pooled_w, float roi_canonical_scale, float roi_canonical_level,
                                     void* workspace, size_t workspace_bytes, void* plan, size_t plan_bytes,
                                     void* stream);
int sd_fpn_roi_align_bwd_packed_plan(const float* out_grad, const float* rois, const uint8_t* argmax,
                                     const float* coords, float* const* d_feats_host, const int* Hs_host,
                                     const int* Ws_host, const int* strides_host, int nlvl, int req_data,
                                     int B, int C, int R, int pooled_h, int pooled_w,
                                     float roi_canonical_scale, float roi_canonical_level, const void* plan,
                                     size_t plan_bytes, void* stream);
/* assign_layer_fpn CustomOp (models/FPN/assign_layer_fpn.py:10-73): rois (n_rois,4) ->
 * rois_per_level (nlvl, n_rois, 4) zero-masked, and optionally level (n_rois) int32 (-1 = none) */
int sd_fpn_roi_assign(const float* rois, int n_rois, const int* strides_host, int nlvl,
                      float roi_canonical_scale, float roi_canonical_level, float* rois_per_level,
                      int32_t* level, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ROIPooling_v1  (mx.sym.ROIPooling_v1)
 *   replaces ROIPoolForward (GPU)  operator_cxx/roi_pooling_v1.cu:48-113 / ROIPoolForward_v1 (CPU)
 *            roi_pooling_v1.cc:39-126, op wrapper roi_pooling_v1-inl.h:70-94
 *   data (B,C,H,W)  rois (K,5) [batch_index,x1,y1,x2,y2]  out, maxidx (K,C,ph,pw)
 *   maxidx holds the flat argmax h*W+w as a float, -1 for an empty bin.
 * ---------------------------------------------------------------------------------------------- */
int sd_roi_pool_v1_fwd(const float* data, const float* rois, float* out, float* maxidx, int B,
                       int C, int H, int W, int K, int pooled_h, int pooled_w, float spatial_scale,
                       void* stream);
/*   replaces ROIPoolBackward  roi_pooling_v1.cu:115-152, wrapper roi_pooling_v1-inl.h:96-133
 *   (d_data honours req_data: write = zero first, add = accumulate; d_rois zeroed on write) */
int sd_roi_pool_v1_bwd(const float* out_grad, const float* rois, const float* maxidx,
                       float* d_data, float* d_rois, int req_data, int req_rois, int B, int C,
                       int H, int W, int K, int pooled_h, int pooled_w, float spatial_scale,
                       void* stream);

/* ------------------------------------------------------------------------------------------------
 * _contrib_GenAnchor  (mx.sym.contrib.GenAnchor)
 *   replaces GenAnchorGPUOp::Forward  operator_cxx/contrib/generate_anchor.cu:97-153
 *            (base anchors generate_anchor-inl.h:140-181 in double on the host, grid kernel :61-81)
 *   out (H*W*A, 4) fp32, A = n_ratios*n_scales (ratio-major), row (h*W + w)*A + a.
 *   The same grid with H = W = max_side/stride is symbol/builder.py:904-938 add_anchor_to_arg.
 * ---------------------------------------------------------------------------------------------- */
int sd_gen_anchor(float* out, int H, int W, int feature_stride, const double* scales_host,
                  int n_scales, const double* ratios_host, int n_ratios, void* stream);
/* the same for every pyramid level in ONE launch (the reference runs one GenAnchor node per level,
 * models/FPN/builder.py; five 2.7 us kernels are launch bound): outs_host[l] = device buffer of
 * level l (Hs[l] * Ws[l] * A rows), same scales / ratios on every level, its own stride */
int sd_gen_anchor_levels(float* const* outs_host, const int* Hs_host, const int* Ws_host,
                         const int* strides_host, int nlvl, const double* scales_host, int n_scales,
                         const double* ratios_host, int n_ratios, void* stream);

/* ------------------------------------------------------------------------------------------------
 * ProposalTarget  (mx.sym.ProposalTarget)
 *   replaces ProposalTargetOp::Forward  operator_cxx/proposal_target-inl.h:123-256 and SampleROI
 *            operator_cxx/proposal_target.cc:21-227 -- which copy to the host and run single
 *            threaded; here everything stays on the device.
 *   rois (B,N,4)  gt_boxes (B,M,5) [x1,y1,x2,y2,cls], cls == -1 padding
 *   roi_output (B,S,4)  label (B,S)  bbox_target, bbox_weight (B,S,4*num_classes)
 *   match_gt_iou (B,S)   (S = image_rois; all outputs are written, kWriteTo semantics)
 *   kept_index (B,S) int32, optional (may be NULL): index into the image's candidate list
 *   [rois with y2 > 0 in order, then the valid gt boxes] of every output row, -1 = unfilled.
 *   rng_state: DEVICE array of 33 int32 = glibc rand() TYPE_3 state (31-word ring, front index,
 *   rear index); advanced in place exactly as the reference advances libc's global state through
 *   std::random_shuffle.  Fill a host copy with sd_glibc_srand_host (seed 1 == never-seeded libc).
 *   workspace: DEVICE scratch of sd_proposal_target_workspace_bytes(B, N, M) bytes.
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int num_classes, batch_images, image_rois;
  float fg_fraction, fg_thresh, bg_thresh_hi, bg_thresh_lo;
  int proposal_without_gt, class_agnostic;
  float bbox_mean[4], bbox_std[4], bbox_weight[4];
} sd_proposal_target_param;
#define SD_GLIBC_RAND_STATE_WORDS 33
int sd_glibc_srand_host(uint32_t seed, int32_t* state_host);
size_t sd_proposal_target_workspace_bytes(int B, int N, int M);
int sd_proposal_target(const float* rois, const float* gt_boxes, int N, int M,
                       const sd_proposal_target_param* param_host, int32_t* rng_state,
                       float* roi_output, float* label, float* bbox_target, float* bbox_weight,
                       float* match_gt_iou, int32_t* kept_index, void* workspace,
                       size_t workspace_bytes, void* stream);
/* ProposalTarget_v2 (mx.sym.ProposalTarget_v2, call site models/tridentnet/builder.py:281-299)
 *   replaces ProposalTargetOp_v2::Forward  operator_cxx/proposal_target_v2-inl.h:128-296 and
 *   proposal_target_v2::SampleROI  operator_cxx/proposal_target_v2.cc:21-177
 *   = ProposalTarget plus valid_ranges (B,2) DEVICE [min, max] object scale per image: with
 *   filter_scales a gt box is appended to the candidate rois only if min^2 <= w*h <= max^2; an
 *   image without candidates / without valid gt gets one all-zero roi / gt row.  image_rois = -1
 *   is rejected (the reference allocates (B,-1,.) tensors for it).  The `ohem` parameter of the
 *   reference is LOG(FATAL) "not implemented" there and has no counterpart here. */
/* ProposalMaskTarget (mx.sym.ProposalMaskTarget, call sites models/maskrcnn/builder.py:115,184,
 * models/tridentnet/builder.py:377,437)
 *   replaces ProposalMaskTargetOp::Forward  operator_cxx/proposal_mask_target-inl.h:141-330,
 *   SampleROIMask proposal_mask_target.cc:218-378 and convertPoly2Mask :148-216 (over rleFrPoly /
 *   rleDecode of the un-vendored COCO mask API)
 *   = the ProposalTarget_v2 sampling (valid_ranges may be NULL: num_args = 3) plus
 *   gt_polys (B,M,L) DEVICE, per gt box [category, n_seg, len_1..len_n, x,y,x,y,...] padded with -1
 *   mask_target (B, FG, mask_size, mask_size), FG = (int)(image_rois * fg_fraction): rows of the
 *   sampled foreground RoIs hold the 0/1 mask of their gt polygon in the RoI's frame, the rest -1.
 *   output_ratio = false; the _ratio entry below is output_ratio = true. */
int sd_proposal_mask_target(const float* rois, const float* gt_boxes, const float* gt_polys,
                            const float* valid_ranges, int filter_scales, int N, int M, int L,
                            int mask_size, const sd_proposal_target_param* param_host,
                            int32_t* rng_state, float* roi_output, float* label, float* bbox_target,
                            float* bbox_weight, float* match_gt_iou, float* mask_target,
                            int32_t* kept_index, void* workspace, size_t workspace_bytes,
                            void* stream);
/* ProposalMaskTarget with output_ratio = true (mask scoring R-CNN, models/msrcnn/builder.py:219-237)
 *   replaces convertPoly2MaskWithRatio  operator_cxx/proposal_mask_target.cc:20-152 (called :368-372)
 *   and the seventh output of ProposalMaskTargetOp  proposal_mask_target-inl.h:244,333-336,453-456
 *   mask_ratio (B, FG): for the sampled foreground rows, the pixels of the gt polygon inside the RoI
 *   (rasterised at image resolution, RoI corners truncated to int) over its pixels inside the
 *   bounding box of RoI and polygon, max(crop / (full + 1e-4), 1e-10); 0 for the other rows.  The
 *   mask itself is computed with the double coordinates of that function (:53-63) -- it can differ
 *   from output_ratio = false in a rounding case, as in the reference.
 *   max_raster_pixels bounds crop_h * crop_w and full_h * full_w (the image area suffices when RoIs
 *   and polygons lie inside the image); a row whose raster is larger gets NaN.
 *   workspace: sd_proposal_mask_target_ratio_workspace_bytes(B, N, M, image_rois, fg_fraction,
 *   max_raster_pixels) bytes (two bitmaps of max_raster_pixels bits per foreground row). */
size_t sd_proposal_mask_target_ratio_workspace_bytes(int B, int N, int M, int image_rois,
                                                     float fg_fraction, int max_raster_pixels);
int sd_proposal_mask_target_ratio(const float* rois, const float* gt_boxes, const float* gt_polys,
                                  const float* valid_ranges, int filter_scales, int N, int M, int L,
                                  int mask_size, const sd_proposal_target_param* param_host,
                                  int32_t* rng_state, float* roi_output, float* label,
                                  float* bbox_target, float* bbox_weight, float* match_gt_iou,
                                  float* mask_target, float* mask_ratio, int max_raster_pixels,
                                  int32_t* kept_index, void* workspace, size_t workspace_bytes,
                                  void* stream);
int sd_proposal_target_v2(const float* rois, const float* gt_boxes, const float* valid_ranges,
                          int filter_scales, int N, int M,
                          const sd_proposal_target_param* param_host, int32_t* rng_state,
                          float* roi_output, float* label, float* bbox_target, float* bbox_weight,
                          float* match_gt_iou, int32_t* kept_index, void* workspace,
                          size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * _contrib_NMS  (mx.sym.contrib.NMS; the same kernel is embedded in Proposal_v3)
 *   replaces NMSGPUOp::Forward  operator_cxx/contrib/nms.cu:249-365 (nms_kernel :102-147, host
 *            scan _nms :149-202 with its D2H/H2D copies, PrepareOutput :207-233)
 *   dets (B,N,5) [x1,y1,x2,y2,score].  pre = pre_nms_top_n > 0 ? min(pre_nms_top_n, N) : N,
 *   post = min(post_nms_top_n, pre).  out (B,post,4), score (B,post): kept boxes in score order,
 *   zero padded.  keep_index (B,post) int32 optional: original row of every kept box, -1 pad.
 *   threshold_ge = 0: suppress IoU > threshold (nms.cu:140); 1: IoU >= threshold
 *   (proposal_v3.cu:319).  The sort is stable (ties keep the lower input row first).
 *   Sizes: pre <= 16384 (one LDS sort; the bit matrix is pre^2 / 8 bytes).  N itself is free
 *   (< 2^24): with more than 16384 unsorted rows the pre best are picked by a radix select and
 *   only they are sorted -- the rows nms.cu:311-313 keeps of its full sort (:303).
 *   workspace: DEVICE scratch of sd_nms_workspace_bytes(B, N, pre_nms_top_n) bytes.
 * ---------------------------------------------------------------------------------------------- */
size_t sd_nms_workspace_bytes(int B, int N, int pre_nms_top_n);
int sd_nms(const float* dets, int B, int N, int pre_nms_top_n, int post_nms_top_n, float threshold,
           int threshold_ge, int already_sorted, float* out, float* score, int32_t* keep_index,
           void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * soft_nms, batched  (operator_py/cython/cpu_nms.pyx:98-203 through operator_py/nms.py:5-16; the
 *   reference runs one Python-object loop per (image, class) in a process pool,
 *   detection_test.py:233-267)
 *   dets (P,Nmax,5): problem p uses rows [0, counts[p]).  method 0 hard, 1 linear, 2 gaussian.
 *   out_dets (P,Nmax,5) / out_inds (P,Nmax): the surviving boxes with their decayed scores and
 *   their input rows, in selection order; out_counts (P).  Rows past out_counts[p] are unspecified.
 * ---------------------------------------------------------------------------------------------- */
int sd_soft_nms_batched(const float* dets, const int32_t* counts, int P, int Nmax, float sigma,
                        float Nt, float threshold, int method, float* out_dets, int32_t* out_inds,
                        int32_t* out_counts, void* stream);
/* bbox_overlaps_cython  (operator_py/cython/bbox.pyx:31-72): overlaps (n,k) */
int sd_bbox_overlaps(const float* boxes, int n, const float* query_boxes, int k, float* overlaps,
                     void* stream);

/* ------------------------------------------------------------------------------------------------
 * RPN anchor-target assignment -- SURVEY 8(f) rank 3.  The reference computes it on the host in its
 * data loader:
 *   replaces AnchorTarget2D.apply  core/detection_input.py:535-565 (anchors :373-437, label
 *            assignment :450-482, random subsampling :484-499, box encoding :501-510, valid-anchor
 *            gather / scatter :512-533) and PyramidAnchorTarget2D.apply  models/FPN/input.py:101-146,
 *            IoU operator_py/cython/bbox.pyx:31-72, encoding operator_py/bbox_transform.py:52-77
 *   im_info (B,3) DEVICE [h, w, scale]; gt_bbox (B,M,G) DEVICE, G = 4 or 5, rows with x1 == -1 padding
 *   mt_state: DEVICE int32[625] = numpy RandomState MT19937 key[624] + position, the generator
 *            np.random.choice draws from; advanced in place exactly as numpy advances it, images in
 *            order.  sd_mt19937_seed_host(seed, ...) fills the state np.random.seed(seed) produces.
 *   layout 0: cls_label (B,N), reg_target / reg_weight (B,N,4) in all-anchor order (level, y, x, a);
 *   layout 1: the loader's final arrays: cls_label (B, A * sumHW), reg_target / reg_weight
 *            (B, 4A, sumHW), levels concatenated along the last axis (one level: (4A, fh, fw)).
 *   An image with h >= w uses the (long, short) feature sizes, otherwise (short, long).
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
  int nlvl;
  int stride[SD_MAX_FPN_LEVELS], short_side[SD_MAX_FPN_LEVELS], long_side[SD_MAX_FPN_LEVELS];
  int n_scales, n_aspects;
  double scales[16], aspects[16]; /* n_scales * n_aspects <= 16 */
  int allowed_border;
  float pos_thr, neg_thr, min_pos_thr;
  int image_anchor;
  double pos_fraction;
} sd_rpn_target_param;
#define SD_MT19937_STATE_WORDS 625
int sd_mt19937_seed_host(uint32_t seed, int32_t* state_host);
int sd_rpn_target_num_anchors(const sd_rpn_target_param* param_host);
size_t sd_rpn_target_workspace_bytes(const sd_rpn_target_param* param_host, int B, int M);
int sd_rpn_anchor_target(const float* im_info, const float* gt_bbox, int B, int M, int G,
                         const sd_rpn_target_param* param_host, int32_t* mt_state,
                         float* cls_label, float* reg_target, float* reg_weight, int layout,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * DeformableConvolution v1  (mx.sym.contrib.DeformableConvolution; call site models/dcn/
 *   builder.py:14-17).  The arithmetic is upstream MXNet 1.6.0 (un-vendored third party,
 *   src/operator/contrib/nn/deformable_im2col.cuh + deformable_convolution-inl.h): each entry point
 *   replaces the function of the same name there.
 *   x (N,C,H,W)  offset (N, dgroup*2*kh*kw, Ho, Wo)  weight (F, C, kh, kw)  y (N,F,Ho,Wo)
 *   col (N, C*kh*kw, Ho*Wo), row (c*kh + i)*kw + j
 * ---------------------------------------------------------------------------------------------- */
int sd_deform_im2col(const float* x, const float* offset, float* col, int N, int C, int H, int W,
                     int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h,
                     int dil_w, int dgroup, void* stream);
/* data gradient: d_x (N,C,H,W) (+)= scatter of col (deformable_col2im) */
int sd_deform_col2im(const float* col, const float* offset, float* d_x, int req, int N, int C,
                     int H, int W, int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w,
                     int dil_h, int dil_w, int dgroup, void* stream);
/* The same with a workspace of sd_deform_col2im_workspace_bytes(N, dgroup) bytes (ABI v7): the gradient planes
 * are summed in 32-bit fixed point with integer LDS adds (2.3x faster than the fp32 compare-and-swap adds of the
 * workspace-free call; bit-reproducible).  The unit is 2^-28 .. 2^-27 of (max|col| of a workgroup's own values
 * x the largest sum of bilinear weights one pixel of its (image, group) can collect); a workgroup whose values
 * are not finite, or whose max|col| x weight bound exceeds 8192 x its mean |col| (heavy-tailed gradients: the
 * unit would be too coarse for the typical element), sums with the float adds instead -- the reference's
 * arithmetic (upstream deformable_col2im: atomicAdd per corner).  This is the path the layer's backward takes. */
size_t sd_deform_col2im_workspace_bytes(int N, int dgroup);
int sd_deform_col2im_ws(const float* col, const float* offset, float* d_x, int req, int N, int C,
                        int H, int W, int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w,
                        int dil_h, int dil_w, int dgroup, void* workspace, size_t workspace_bytes,
                        void* stream);
/* offset gradient: d_offset like offset (deformable_col2im_coord) */
int sd_deform_col2im_coord(const float* col, const float* x, const float* offset, float* d_offset,
                           int req, int N, int C, int H, int W, int kh, int kw, int pad_h,
                           int pad_w, int stride_h, int stride_w, int dil_h, int dil_w, int dgroup,
                           void* stream);
/* fp32-in / fp32-out matrix-core GEMM (row-major, batched): C[b] = op(A[b]) . op(B[b]); op(A) is
 * M x K.  accumulate: 0 store, 1 C += product, 2 atomic add (several batches into one C:
 * strideC = 0).  Replaces the linalg_gemm calls of deformable_convolution-inl.h (cuBLAS sgemm in
 * the reference).
 * Arithmetic (tuning key `deform_gemm_split`):
 *   2 (default)  scaled fp16 split: a pre-pass takes max|A| and max|B|, each operand is scaled by the
 *      power of two that brings its maximum into [2^13, 2^14) and split into two fp16 parts
 *      (hi = RNE(x s), lo = RNE(x s - hi): 22 mantissa bits), a product is a_hi*b_hi + a_hi*b_lo +
 *      a_lo*b_hi on v_mfma_f32_32x32x16_f16 (hi*hi exact in the fp32 accumulator), the result is scaled
 *      back exactly.  Error against fp64 = that of the fp32 MFMA path (measured 5e-7 x max|C| at K =
 *      2304; tests hold it to <= 2x the exact path's).  Elements below 2^-17 of their operand's maximum
 *      lose relative (not absolute) precision; non-finite inputs give NaN.  Needs the maxima, i.e. a
 *      workspace: sd_gemm_f32_ws and the sd_deform_conv_* entry points; plain sd_gemm_f32 runs the exact
 *      path (0) instead.
 *   1  bf16 split, no pre-pass: hi = RNE(x), lo = RNE(x - hi) in bf16, 16 mantissa bits: 4.5e-6 x
 *      max|C| (nine times the exact path's error); opt-in.
 *   0  v_mfma_f32_32x32x2_f32: exact fp32 products, 5e-7 x max|C|, ~2.5x slower.
 *   Split paths: when the last round of resident workgroups would be under half full its tiles are cut
 *   into k slices that add atomically (those tiles' last bits then depend on the order; key
 *   `deform_gemm_ksplit` = 0 turns that off). */
int sd_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, int lda, long strideA,
                const float* B, int ldb, long strideB, float* C, int ldc, long strideC, int batch,
                int accumulate, void* stream);
/* the same with a workspace of sd_gemm_f32_workspace_bytes() bytes for the operand maxima (two extra
 * small launches that read A and B once) */
size_t sd_gemm_f32_workspace_bytes(void);
int sd_gemm_f32_ws(int transA, int transB, int M, int N, int K, const float* A, int lda, long strideA,
                   const float* B, int ldb, long strideB, float* C, int ldc, long strideC, int batch,
                   int accumulate, void* workspace, size_t workspace_bytes, void* stream);
size_t sd_deform_conv_workspace_bytes(int N, int C, int H, int W, int kh, int kw, int pad,
                                      int stride, int dil);
/* forward WITHOUT the col matrix: deformable sampling fused into the GEMM's B-operand staging (every
 * sample taken once, all filters of a pixel tile in one workgroup).  3x3 kernels, C / dgroup % 16 == 0
 * and H*W % 4 == 0 (every DCN layer of the reference's configs); other shapes run the unfused forward
 * below through the same call (the workspace size says which: a few MB of pre-split weights against
 * the N*C*9*Ho*Wo*4-byte col matrix).  Same arithmetic as the unfused path (sampled values bit-equal to
 * sd_deform_im2col's, products by the scaled fp16 split), another summation order over k.
 *   replaces DeformableConvolutionOp::Forward (upstream MXNet 1.6 deformable_convolution-inl.h;
 *   call site models/dcn/builder.py:14-17) */
size_t sd_deform_conv_fwd_nocol_workspace_bytes(int N, int C, int H, int W, int F, int kh, int kw, int pad,
                                                int stride, int dil, int dgroup);
int sd_deform_conv_fwd_nocol(const float* x, const float* offset, const float* weight, float* y, int N,
                             int C, int H, int W, int F, int kh, int kw, int pad, int stride, int dil,
                             int dgroup, void* workspace, size_t workspace_bytes, void* stream);
/* forward = im2col + GEMM, num_group = 1, no bias (the reference's configuration) */
int sd_deform_conv_fwd(const float* x, const float* offset, const float* weight, float* y, int N,
                       int C, int H, int W, int F, int kh, int kw, int pad, int stride, int dil,
                       int dgroup, void* workspace, size_t workspace_bytes, void* stream);
/* backward: d_x, d_offset, d_weight with their own req (0 null / 1 write / 3 add) */
int sd_deform_conv_bwd(const float* out_grad, const float* x, const float* offset,
                       const float* weight, float* d_x, float* d_offset, float* d_weight,
                       int req_x, int req_offset, int req_weight, int N, int C, int H, int W,
                       int F, int kh, int kw, int pad, int stride, int dil, int dgroup,
                       void* workspace, size_t workspace_bytes, void* stream);
/* The same backward with the col matrix of the forward kept instead of recomputed: fwd_col =
 * sd_deform_conv_col_of_workspace(the workspace sd_deform_conv_fwd ran with, untouched since); the
 * backward needs a workspace of its own (dcol).  Same results as sd_deform_conv_bwd (im2col is
 * deterministic: the col matrix is the same bits); it trades N*C*kh*kw*Ho*Wo*4 bytes held per layer between the two
 * calls for the im2col pass (0.29 of 1.85 ms on the (16,256,50,84) layer). */
const float* sd_deform_conv_col_of_workspace(const void* fwd_workspace);
int sd_deform_conv_bwd_cached(const float* out_grad, const float* x, const float* offset,
                              const float* weight, const float* fwd_col, float* d_x,
                              float* d_offset, float* d_weight, int req_x, int req_offset,
                              int req_weight, int N, int C, int H, int W, int F, int kh, int kw,
                              int pad, int stride, int dil, int dgroup, void* workspace,
                              size_t workspace_bytes, void* stream);

/* The operator with ALL of DeformableConvolutionParam (upstream MXNet 1.6 deformable_convolution-inl.h:
 * kernel, stride, dilate, pad, num_filter, num_group, num_deformable_group, no_bias) -- the call sites
 * beside models/dcn/builder.py pass a bias (models/RepPoints/builder.py:215-245, no_bias=False) and
 * num_group / bias through (models/sepc/sepc_dconv.py:5-16; models/tridentnet/resnet_v1.py:85-90).
 *   weight (F, C / num_group, kh, kw): filter block g (F / num_group filters) sees input channels
 *   [g C / num_group, (g + 1) C / num_group) (the col rows of those channels), as the reference's loop
 *   over group_ does; bias (F) or NULL (= no_bias): y[n, f, :] += bias[f] after the products, as
 *   `out += broadcast<1>(bias)` does.  d_bias (F) (+)= sum over n, pixels of out_grad
 *   (`sumall_except_dim<1>`), its own req.
 * forward: keep_col = 0 -> the col-free fused kernel where the shape allows (num_group = 1, see
 *   sd_deform_conv_fwd_nocol; the bias is added in its epilogue), else im2col + one GEMM per group + a
 *   bias pass; keep_col = 1 -> always the latter, and the col matrix stays in the workspace for
 *   sd_deform_convolution_bwd(fwd_col = sd_deform_conv_col_of_workspace(workspace)).
 * backward: fwd_col NULL -> recomputed.  Workspace: sd_deform_conv_workspace_bytes. */
size_t sd_deform_convolution_fwd_workspace_bytes(int N, int C, int H, int W, int F, int kh, int kw, int pad,
                                                 int stride, int dil, int dgroup, int num_group, int keep_col);
int sd_deform_convolution_fwd(const float* x, const float* offset, const float* weight, const float* bias,
                              float* y, int N, int C, int H, int W, int F, int kh, int kw, int pad, int stride,
                              int dil, int dgroup, int num_group, int keep_col, void* workspace,
                              size_t workspace_bytes, void* stream);
int sd_deform_convolution_bwd(const float* out_grad, const float* x, const float* offset, const float* weight,
                              const float* fwd_col, float* d_x, float* d_offset, float* d_weight, float* d_bias,
                              int req_x, int req_offset, int req_weight, int req_bias, int N, int C, int H,
                              int W, int F, int kh, int kw, int pad, int stride, int dil, int dgroup,
                              int num_group, void* workspace, size_t workspace_bytes, void* stream);

/* ------------------------------------------------------------------------------------------------
 * _contrib_Proposal_v3  (mx.sym.contrib.Proposal_v3, models/FPN/builder.py:275-287) -- SURVEY 8(f)
 *   replaces ProposalGPUOp_v3::Forward  operator_cxx/contrib/proposal_v3.cu:428-638 (anchor grid
 *   :64-85, box decode :92-155, top-k by thrust sort, min-size filter :211-235, NMS with >= and its
 *   host scan :271-381, PrepareOutput :386-416; three D2H/H2D copies per image)
 *   cls_prob (B,2A,H,W) (foreground = second half)  bbox_pred (B,4A,H,W)  im_info (B,3) DEVICE
 *   out (B,post,4)  score (B,post): post = is_train ? min(post_nms_top_n, pre) : post_nms_top_n;
 *   padding past the kept boxes: zeros (test) or the kept boxes repeated cyclically (is_train).
 *   sd_proposal_v3_iou is the op with iou_loss = true: IoUPredKernel (proposal_v3.cu:163-205, the
 *   four deltas are added to the anchor's corners) in place of BBoxPredKernel; no reference config
 *   enables it.
 * ---------------------------------------------------------------------------------------------- */
size_t sd_proposal_v3_workspace_bytes(int B, int A, int H, int W, int pre_nms_top_n);
int sd_proposal_v3(const float* cls_prob, const float* bbox_pred, const float* im_info, float* out,
                   float* score, int B, int A, int H, int W, int rpn_pre_nms_top_n,
                   int rpn_post_nms_top_n, float threshold, int rpn_min_size,
                   const float* scales_host, int n_scales, const float* ratios_host, int n_ratios,
                   int feature_stride, int is_train, void* workspace, size_t workspace_bytes,
                   void* stream);
int sd_proposal_v3_iou(const float* cls_prob, const float* bbox_pred, const float* im_info,
                       float* out, float* score, int B, int A, int H, int W, int rpn_pre_nms_top_n,
                       int rpn_post_nms_top_n, float threshold, int rpn_min_size,
                       const float* scales_host, int n_scales, const float* ratios_host,
                       int n_ratios, int feature_stride, int is_train, void* workspace,
                       size_t workspace_bytes, void* stream);
/* get_top_proposal CustomOp (models/FPN/get_top_proposal.py:15-39): the top_n rows of bbox (B,N,4)
 * by score (B,N) descending (ties: lower row first), zero padded when N < top_n */
int sd_get_top_proposal(const float* bbox, const float* score, int B, int N, int top_n,
                        float* out_bbox, float* out_score, void* stream);

/* ------------------------------------------------------------------------------------------------
 * _contrib_DecodeBBox  (mx.sym.contrib.DecodeBBox / X.decode_bbox, symbol/builder.py:384-392) and
 * the test-time per-class filter in front of soft-NMS -- SURVEY 8(f) rank 2
 *   replaces DecodeBBoxOp::Forward  operator_cxx/contrib/decodebbox.cc:147-215 (host round trip +
 *   BBoxTransformXYWH :34-80 / BBoxTransformXYXY :84-131)
 *   rois (B,R,4)  bbox_pred (B,R,4K)  im_info (B,3) DEVICE  out (B,R,4K), or (B,R,4) when
 *   class_agnostic (then the deltas of class 1 are used, decodebbox.cc:56)
 * ---------------------------------------------------------------------------------------------- */
int sd_decode_bbox(const float* rois, const float* bbox_pred, const float* im_info, float* out,
                   int B, int R, int K, const float* bbox_mean_host, const float* bbox_std_host,
                   int class_agnostic, int decode_xyxy, void* stream);
/* detection_test.py:233-247 (do_nms): for every (image, class) the rows with
 * cls_score > min_det_score as [x1,y1,x2,y2,score], row order kept -- written in the layout
 * sd_soft_nms_batched reads: dets (B*K, R, 5), counts (B*K).  bbox (B,R,4*bbox_classes),
 * bbox_classes = K (class specific boxes) or 1 (shared box). */
int sd_det_filter(const float* bbox, const float* cls_score, int B, int R, int K, int bbox_classes,
                  float min_det_score, float* dets, int32_t* counts, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SIMPLEDET_OPS_H_ */
