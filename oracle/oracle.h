/*
 * oracle.h -- CPU restatement ("oracle") of the SimpleDet second-stage detection ops.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under simpledet_amd/ may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the checker
 * (or as the reported-only CPU baseline), never as the product path.
 *
 * Every function follows the reference file:line cited above it (paths relative to the
 * tusen-ai/simpledet tree).  The reference's own operator_cxx/ sources need MXNet/mshadow/dmlc
 * headers that are not available, so they cannot be compiled here; this is a restatement.
 * Pins (see oracle/README.md): the ROIPooling docstring golden vector, the reference's own
 * Cython soft_nms / greedy_nms / bbox_overlaps built into oracle/_ref, numpy twins imported from
 * the reference (anchors, assign_layer_fpn, py nms) -> tests/golden/, libstdc++ random_shuffle.
 *
 * Build: gcc -O2 -ffp-contract=off (strict IEEE float, no FMA contraction, no fast-math) so that
 * float results are reproducible bit-for-bit by a device kernel built the same way.
 */
#ifndef SIMPLEDET_ORACLE_H_
#define SIMPLEDET_ORACLE_H_
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ---- RoIAlign_v2 : operator_cxx/contrib/roi_align_v2-inl.h:61-153, roi_align_v2.cu:35-84 ---- */
void orc_roi_align_v2_fwd(const float* data, const float* rois, float* out, float* amax_x,
                          float* amax_y, int B, int C, int H, int W, int R, int ph, int pw,
                          float spatial_scale, int nthreads);
/* GPU (scatter) semantics: roi_align_v2.cu:35-84.  req: 1 = write (zero first), 3 = add. */
void orc_roi_align_v2_bwd(const float* dy, const float* amax_x, const float* amax_y, float* dx,
                          int B, int C, int H, int W, int R, int ph, int pw, int req, int nthreads);
/* CPU (gather) semantics: roi_align_v2.cc:35-106 -- differs on degenerate bins. */
void orc_roi_align_v2_bwd_cpu_gather(const float* dy, const float* rois, const float* amax_x,
                                     const float* amax_y, float* dx, int B, int C, int H, int W,
                                     int R, int ph, int pw, float spatial_scale, int req);
/* FPN level assignment: models/FPN/assign_layer_fpn.py:17-41.  level[i] in {0..nlvl-1} or -1. */
void orc_fpn_roi_assign(const float* rois, int n_rois, const int* strides, int nlvl,
                        float canonical_scale, float canonical_level, int* level,
                        float* rois_per_level /* nlvl x n_rois x 4, may be NULL */);
/* Reference graph models/FPN/builder.py:563-610: assign -> per level ROIAlign_v2 -> add_n. */
void orc_fpn_roi_align_fwd(const float* const* feats, const int* Hs, const int* Ws,
                           const int* strides, int nlvl, const float* rois, float* out,
                           float* amax_x, float* amax_y, int B, int C, int R, int ph, int pw,
                           float canonical_scale, float canonical_level, int nthreads);
void orc_fpn_roi_align_bwd(const float* dy, const float* rois, const float* amax_x,
                           const float* amax_y, float* const* dfeats, const int* Hs,
                           const int* Ws, const int* strides, int nlvl, int B, int C, int R,
                           int ph, int pw, float canonical_scale, float canonical_level, int req,
                           int nthreads);

/* ---- ROIPooling_v1 : operator_cxx/roi_pooling_v1.cc:39-221, roi_pooling_v1.cu:48-152 ---- */
void orc_roi_pool_v1_fwd(const float* data, const float* rois, float* out, float* maxidx, int B,
                         int C, int H, int W, int K, int ph, int pw, float spatial_scale);
void orc_roi_pool_v1_bwd(const float* dy, const float* rois, const float* maxidx, float* dx,
                         int B, int C, int H, int W, int K, int ph, int pw, float spatial_scale,
                         int req);
/* CPU gather form roi_pooling_v1.cc:128-221 (same result as the scatter on valid argmax). */
void orc_roi_pool_v1_bwd_cpu_gather(const float* dy, const float* rois, const float* maxidx,
                                    float* dx, int B, int C, int H, int W, int K, int ph, int pw,
                                    float spatial_scale, int req);

/* ---- GenAnchor : contrib/generate_anchor-inl.h:140-181, generate_anchor.cc:38-83 ---- */
void orc_gen_base_anchors(int feature_stride, const double* scales, int ns, const double* ratios,
                          int nr, double* base /* nr*ns*4 */);
void orc_gen_anchor(float* out, int H, int W, int feature_stride, const double* scales, int ns,
                    const double* ratios, int nr);

/* ---- ProposalTarget : operator_cxx/proposal_target-inl.h:123-256, proposal_target.cc:21-227 -- */
typedef struct {
  int num_classes, batch_images, image_rois;
  float fg_fraction, fg_thresh, bg_thresh_hi, bg_thresh_lo;
  int proposal_without_gt, class_agnostic;
  float bbox_mean[4], bbox_std[4], bbox_weight[4];
} orc_proposal_target_param;
/* glibc TYPE_3 rand() restatement (r[i] = r[i-3] + r[i-31]); state = 34 words + 2 indices. */
typedef struct { int32_t r[34]; int32_t f, b; } orc_glibc_rand;
void orc_glibc_srand(orc_glibc_rand* st, unsigned seed);
int orc_glibc_rand_next(orc_glibc_rand* st);
/* returns 0, or -1 when the reference would read out of bounds (an image without a valid gt) */
int orc_proposal_target(const float* rois, const float* gt_boxes, int N, int M,
                        const orc_proposal_target_param* p, orc_glibc_rand* rng,
                        float* roi_out, float* label, float* bbox_target, float* bbox_weight,
                        float* match_gt_iou, int* kept_index /* B*S, may be NULL; -1 = unfilled */);
/* same, but driving libc rand() itself (what the reference binary calls) */
/* ProposalTarget_v2 (proposal_target_v2-inl.h / .cc): valid_ranges (B,2), filter_scales */
int orc_proposal_target_v2(const float* rois, const float* gt_boxes, const float* valid_ranges,
                           int filter_scales, int N, int M, const orc_proposal_target_param* p,
                           orc_glibc_rand* rng, float* roi_out, float* label, float* bbox_target,
                           float* bbox_weight, float* match_gt_iou, int* kept_index);
/* ProposalMaskTarget (proposal_mask_target-inl.h / .cc), output_ratio = false (the _ratio variant below: true) */
int orc_proposal_mask_target(const float* rois, const float* gt_boxes, const float* gt_polys,
                             const float* valid_ranges, int filter_scales, int N, int M, int L,
                             int mask_size, const orc_proposal_target_param* p, orc_glibc_rand* rng,
                             float* roi_out, float* label, float* bbox_target, float* bbox_weight,
                             float* match_gt_iou, int* kept_index, float* mask_target);
int orc_poly2mask(const float* roi, const float* poly, int mask_size, float* mask);
/* output_ratio = true (proposal_mask_target.cc:20-152, 368-372): + mask_ratio (B, FG) */
int orc_proposal_mask_target_ratio(const float* rois, const float* gt_boxes, const float* gt_polys,
                                   const float* valid_ranges, int filter_scales, int N, int M, int L,
                                   int mask_size, const orc_proposal_target_param* p,
                                   orc_glibc_rand* rng, float* roi_out, float* label,
                                   float* bbox_target, float* bbox_weight, float* match_gt_iou,
                                   int* kept_index, float* mask_target, float* mask_ratio);
int orc_poly2mask_ratio(const float* roi, const float* poly, int mask_size, float* mask, double* ratio);
int orc_proposal_target_libc(const float* rois, const float* gt_boxes, int N, int M,
                             const orc_proposal_target_param* p, float* roi_out, float* label,
                             float* bbox_target, float* bbox_weight, float* match_gt_iou,
                             int* kept_index);

/* ---- _contrib_NMS (GPU path is the spec): contrib/nms.cu:92-202,207-233,249-365 ---- */
void orc_nms(const float* dets, int B, int N, int pre_nms_top_n, int post_nms_top_n,
             float threshold, int already_sorted, float* out, float* score,
             int* keep_out /* B*post, original indices, -1 pad; may be NULL */);

/* ---- _contrib_Proposal_v3 (GPU path): contrib/proposal_v3.cu:64-235,271-416,428-638 ---- */
void orc_proposal_v3_anchors(int feature_stride, const float* scales, int ns, const float* ratios,
                             int nr, float* anchors);
int orc_proposal_v3_post(int count, int pre_nms_top_n, int post_nms_top_n, int is_train);
void orc_proposal_v3(const float* cls_prob, const float* bbox_pred, const float* im_info, int B,
                     int A, int H, int W, int pre_nms_top_n, int post_nms_top_n, float threshold,
                     int min_size, const float* scales, int ns, const float* ratios, int nr,
                     int feature_stride, int is_train, float* out, float* score_out);
/* iou_loss = true: IoUPredKernel (proposal_v3.cu:163-205) instead of BBoxPredKernel */
void orc_proposal_v3_iou(const float* cls_prob, const float* bbox_pred, const float* im_info, int B,
                         int A, int H, int W, int pre_nms_top_n, int post_nms_top_n, float threshold,
                         int min_size, const float* scales, int ns, const float* ratios, int nr,
                         int feature_stride, int is_train, float* out, float* score_out);
/* models/FPN/get_top_proposal.py:15-39 */
void orc_get_top_proposal(const float* bbox, const float* score, int B, int N, int top_n,
                          float* out_bbox, float* out_score);

/* ---- _contrib_DecodeBBox: contrib/decodebbox.cc:34-131; test-time filter detection_test.py:233-247 */
void orc_decode_bbox(const float* rois, const float* deltas, const float* im_info, float* out,
                     int B, int R, int K, const float* means, const float* stds,
                     int class_agnostic, int xyxy);
void orc_det_filter(const float* bbox, const float* cls_score, int B, int R, int K, int Kb,
                    float min_det_score, float* dets, int* counts);

/* ---- soft_nms : operator_py/cython/cpu_nms.pyx:98-203 ---- */
/* boxes (n,5) is updated in place (as the Cython copy is); returns new N; inds (n) out. */
int orc_soft_nms(float* boxes, int64_t* inds, int n, float sigma, float Nt, float threshold,
                 unsigned method);
/* greedy_nms cpu_nms.pyx:37-87 (>= thresh): keep mask (n) out, returns count */
int orc_greedy_nms(const float* dets, int n, float thresh, int64_t* keep);
/* bbox_overlaps_cython operator_py/cython/bbox.pyx:31-72 */
void orc_bbox_overlaps(const float* boxes, int n, const float* query, int k, float* overlaps);

/* ---- Deformable convolution v1 (apache/incubator-mxnet 1.6.0, un-vendored: parity unpinned) -- */
void orc_deform_im2col(const float* x, const float* offset, float* col, int C, int H, int W,
                       int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w,
                       int dil_h, int dil_w, int dgroup, int Ho, int Wo);
void orc_deform_col2im(const float* col, const float* offset, float* dx, int C, int H, int W,
                       int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w,
                       int dil_h, int dil_w, int dgroup, int Ho, int Wo);
void orc_deform_col2im_coord(const float* col, const float* x, const float* offset, float* doff,
                             int C, int H, int W, int kh, int kw, int pad_h, int pad_w,
                             int stride_h, int stride_w, int dil_h, int dil_w, int dgroup, int Ho,
                             int Wo);
/* y(F,Ho*Wo) = Wt(F, C*kh*kw) . col  (group=1), fp32 accumulate in k order */
void orc_deform_conv_fwd(const float* x, const float* offset, const float* wt, float* y, int N,
                         int C, int H, int W, int F, int kh, int kw, int pad, int stride, int dil,
                         int dgroup);
void orc_deform_convolution_fwd(const float* x, const float* offset, const float* wt, const float* bias,
                                float* y, int N, int C, int H, int W, int F, int kh, int kw, int pad,
                                int stride, int dil, int dgroup, int num_group);
void orc_deform_convolution_bwd(const float* dy, const float* x, const float* offset, const float* wt,
                                float* dx, float* doff, float* dw, float* dbias, int N, int C, int H, int W,
                                int F, int kh, int kw, int pad, int stride, int dil, int dgroup,
                                int num_group);

#ifdef __cplusplus
}
#endif
#endif
