/*
 * ORACLE (test infrastructure, not product code) -- CPU restatement of ROIAlign_v2.
 *
 * Follows, expression by expression and type by type (DType = float):
 *   forward   operator_cxx/contrib/roi_align_v2-inl.h:61-153  (ROIAlignForwardKernel_v2::Map)
 *   backward  operator_cxx/contrib/roi_align_v2.cu:35-84      (ROIAlignBackwardKernelGPU_v2::Map,
 *             scatter semantics = the spec; launcher zero-fill/add :130-137)
 *   backward' operator_cxx/contrib/roi_align_v2.cc:35-106     (ROIAlignBackwardKernelCPU::Map,
 *             gather semantics, kept only to demonstrate the documented divergence)
 *   assign    models/FPN/assign_layer_fpn.py:17-41            (AssignLayerFPNOperator.forward)
 *   graph     models/FPN/builder.py:563-610                   (FPNRoiAlign.get_roi_feature)
 *
 * Mixed float/double expressions of the reference are kept: "/3.0" and "+0.01" are double
 * (roi_align_v2-inl.h:120-125).  Compile with -ffp-contract=off.
 */
#include "oracle.h"
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline float fmaxr(float a, float b) { return a > b ? a : b; } /* mshadow_op::maximum */
static inline float fminr(float a, float b) { return a < b ? a : b; } /* mshadow_op::minimum */
static inline int imaxr(int a, int b) { return a > b ? a : b; }
static inline int iminr(int a, int b) { return a < b ? a : b; }

/* one output element; roi_align_v2-inl.h:70-153 */
static void fwd_elem(long index, const float* bottom_data, int num_rois_per_batch,
                     float spatial_scale, int channels, int height, int width, int pooled_height,
                     int pooled_width, const float* bottom_rois, float* top_data, float* argmax_x,
                     float* argmax_y) {
  int pw = (int)(index % pooled_width);
  int ph = (int)((index / pooled_width) % pooled_height);
  int c = (int)((index / pooled_width / pooled_height) % channels);
  int n = (int)(index / pooled_width / pooled_height / channels);

  bottom_rois += (long)n * 4;
  int roi_batch_ind = n / num_rois_per_batch; /* :77 ; the <0 branch :79-84 is dead code */

  float roi_start_w = bottom_rois[0] * spatial_scale;
  float roi_start_h = bottom_rois[1] * spatial_scale;
  float roi_end_w = bottom_rois[2] * spatial_scale;
  float roi_end_h = bottom_rois[3] * spatial_scale;
  float roi_width = roi_end_w - roi_start_w;
  float roi_height = roi_end_h - roi_start_h;
  float bin_size_h = roi_height / (float)pooled_height;
  float bin_size_w = roi_width / (float)pooled_width;

  float hstart = (float)(ph)*bin_size_h;
  float wstart = (float)(pw)*bin_size_w;
  float hend = (float)(ph + 1) * bin_size_h;
  float wend = (float)(pw + 1) * bin_size_w;
  hstart = fminr(fmaxr(hstart + roi_start_h, 0.f), (float)(height - 1));
  hend = fminr(fmaxr(hend + roi_start_h, 0.f), (float)(height - 1));
  wstart = fminr(fmaxr(wstart + roi_start_w, 0.f), (float)(width - 1));
  wend = fminr(fmaxr(wend + roi_start_w, 0.f), (float)(width - 1));
  int is_empty = (hend <= hstart) || (wend <= wstart);

  float maxidx_x = -1.f, maxidx_y = -1.f, maxval = 0.f;
  if (!is_empty) {
    maxval = -FLT_MAX; /* mshadow::red::limits::MinValue<float>() */
    bottom_data += ((long)roi_batch_ind * channels + c) * height * width;
    float h_stride = (float)((double)(hend - hstart) / 3.0);
    float w_stride = (float)((double)(wend - wstart) / 3.0);
    for (float h = hstart + h_stride; (double)h <= (double)(hend - h_stride) + 0.01;
         h += fmaxr(h_stride, 0.01f)) {
      for (float w = wstart + w_stride; (double)w <= (double)(wend - w_stride) + 0.01;
           w += fmaxr(w_stride, 0.01f)) {
        int hlow = iminr(imaxr((int)floorf(h), 0), height - 1);
        int hhigh = iminr(imaxr((int)ceilf(h), 0), height - 1);
        int wleft = iminr(imaxr((int)floorf(w), 0), width - 1);
        int wright = iminr(imaxr((int)ceilf(w), 0), width - 1);
        int topleft = hlow * width + wleft;
        int topright = hlow * width + wright;
        int bottomleft = hhigh * width + wleft;
        int bottomright = hhigh * width + wright;
        float alpha = (hlow == hhigh) ? 0.5f : (h - (float)hlow) / (float)(hhigh - hlow);
        float beta = (wleft == wright) ? 0.5f : (w - (float)wleft) / (float)(wright - wleft);
        float value = (1 - alpha) * (1 - beta) * bottom_data[topleft] +
                      alpha * (1 - beta) * bottom_data[bottomleft] +
                      (1 - alpha) * beta * bottom_data[topright] +
                      alpha * beta * bottom_data[bottomright];
        if (value > maxval) {
          maxval = value;
          maxidx_x = w;
          maxidx_y = h;
        }
      }
    }
  }
  top_data[index] = maxval;
  argmax_x[index] = maxidx_x;
  argmax_y[index] = maxidx_y;
}

void orc_roi_align_v2_fwd(const float* data, const float* rois, float* out, float* amax_x,
                          float* amax_y, int B, int C, int H, int W, int R, int ph, int pw,
                          float spatial_scale, int nthreads) {
  long count = (long)B * R * C * ph * pw;
  (void)nthreads;
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
  for (long i = 0; i < count; ++i)
    fwd_elem(i, data, R, spatial_scale, C, H, W, ph, pw, rois, out, amax_x, amax_y);
}

/* roi_align_v2.cu:35-84, launcher :130-137 (Fill 0 when kWriteTo).
 * The reference is one atomicAdd per tap in arbitrary order; here taps are applied plane by plane
 * (image b, channel c) in ascending (roi, ph, pw) order, which is also the serial index order of
 * the reference restricted to that plane -- so nthreads only changes speed, never the result. */
void orc_roi_align_v2_bwd(const float* dy, const float* amax_x, const float* amax_y, float* dx,
                          int B, int C, int H, int W, int R, int ph, int pw, int req,
                          int nthreads) {
  if (req == 0) return;
  if (req == 1) memset(dx, 0, sizeof(float) * (size_t)B * C * H * W);
  const int PP = ph * pw;
  (void)nthreads;
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
  for (long bc = 0; bc < (long)B * C; ++bc) {
    int b = (int)(bc / C), c = (int)(bc % C);
    float* offset_bottom_diff = dx + bc * H * W;
    for (int r = 0; r < R; ++r) {
      long base = (((long)b * R + r) * C + c) * PP;
      for (int p = 0; p < PP; ++p) {
        float a_x = amax_x[base + p];
        float a_y = amax_y[base + p];
        if (a_x != -1.f && a_y != -1.f) {
          int hlow = iminr(imaxr((int)floorf(a_y), 0), H - 1);
          int hhigh = iminr(imaxr((int)ceilf(a_y), 0), H - 1);
          int wleft = iminr(imaxr((int)floorf(a_x), 0), W - 1);
          int wright = iminr(imaxr((int)ceilf(a_x), 0), W - 1);
          float alpha = (hlow == hhigh) ? 0.5f : (a_y - (float)hlow) / (float)(hhigh - hlow);
          float beta = (wleft == wright) ? 0.5f : (a_x - (float)wleft) / (float)(wright - wleft);
          float g = dy[base + p];
          offset_bottom_diff[hlow * W + wleft] += g * (1 - alpha) * (1 - beta);
          offset_bottom_diff[hlow * W + wright] += g * (1 - alpha) * beta;
          offset_bottom_diff[hhigh * W + wleft] += g * alpha * (1 - beta);
          offset_bottom_diff[hhigh * W + wright] += g * alpha * beta;
        }
      }
    }
  }
}

/* roi_align_v2.cc:35-106 (the reference's CPU backward; NOT the spec) */
void orc_roi_align_v2_bwd_cpu_gather(const float* dy, const float* rois, const float* amax_x,
                                     const float* amax_y, float* dx, int B, int C, int H, int W,
                                     int R, int ph, int pw, float spatial_scale, int req) {
  if (req == 0) return;
  if (req == 1) memset(dx, 0, sizeof(float) * (size_t)B * C * H * W);
  long count = (long)B * C * H * W;
  int num_rois = B * R;
  for (long index = 0; index < count; ++index) {
    int w = (int)(index % W);
    int h = (int)((index / W) % H);
    int c = (int)((index / W / H) % C);
    int n = (int)(index / W / H / C);
    float gradient = 0;
    for (int roi_n = 0; roi_n < num_rois; ++roi_n) {
      const float* r = rois + (long)roi_n * 4;
      if (n != roi_n / R) continue;
      float roi_start_w = r[0] * spatial_scale;
      float roi_start_h = r[1] * spatial_scale;
      float roi_end_w = r[2] * spatial_scale;
      float roi_end_h = r[3] * spatial_scale;
      int in_roi = ((double)w > (double)roi_start_w - 1.0 && (double)w < (double)roi_end_w + 1.0 &&
                    (double)h > (double)roi_start_h - 1.0 && (double)h < (double)roi_end_h + 1.0);
      if (!in_roi) continue;
      long offset = ((long)roi_n * C + c) * ph * pw;
      for (int p = 0; p < ph * pw; ++p) {
        float a_x = amax_x[offset + p];
        float a_y = amax_y[offset + p];
        int hlow = iminr(imaxr((int)floorf(a_y), 0), H - 1);
        int hhigh = iminr(imaxr((int)ceilf(a_y), 0), H - 1);
        int wleft = iminr(imaxr((int)floorf(a_x), 0), W - 1);
        int wright = iminr(imaxr((int)ceilf(a_x), 0), W - 1);
        if (h != hlow && h != hhigh && w != wleft && w != wright) continue;
        float alpha = (hlow == hhigh) ? 0.5f : (a_y - (float)hlow) / (float)(hhigh - hlow);
        float beta = (wleft == wright) ? 0.5f : (a_x - (float)wleft) / (float)(wright - wleft);
        float g = dy[offset + p];
        if (h == hlow && w == wleft) gradient += g * (1 - alpha) * (1 - beta);
        else if (h == hlow && w == wright) gradient += g * (1 - alpha) * beta;
        else if (h == hhigh && w == wleft) gradient += g * alpha * (1 - beta);
        else if (h == hhigh && w == wright) gradient += g * alpha * beta;
      }
    }
    dx[index] += gradient;
  }
}

/* models/FPN/assign_layer_fpn.py:17-41, all in float32 as mx.nd evaluates it */
void orc_fpn_roi_assign(const float* rois, int n_rois, const int* strides, int nlvl,
                        float canonical_scale, float canonical_level, int* level,
                        float* rois_per_level) {
  int smin = strides[0], smax = strides[0];
  for (int l = 1; l < nlvl; ++l) {
    if (strides[l] < smin) smin = strides[l];
    if (strides[l] > smax) smax = strides[l];
  }
  float k_min = (float)log2((double)smin), k_max = (float)log2((double)smax);
  if (rois_per_level) memset(rois_per_level, 0, sizeof(float) * (size_t)nlvl * n_rois * 4);
  for (int i = 0; i < n_rois; ++i) {
    const float* r = rois + (long)i * 4;
    float area = (r[2] - r[0] + 1.f) * (r[3] - r[1] + 1.f);
    float scale = sqrtf(area);
    float t = floorf(canonical_level + log2f(scale / canonical_scale + 1e-6f));
    /* mx.nd.clip: x < a_min ? a_min : (x > a_max ? a_max : x); NaN passes through */
    t = t < k_min ? k_min : (t > k_max ? k_max : t);
    float ts = powf(2.f, t);
    /* .astype('uint8') then "== s"; NaN -> matches no stride */
    int lvl = -1;
    if (ts == ts) {
      unsigned char u8 = (unsigned char)(int)ts;
      for (int l = 0; l < nlvl; ++l)
        if ((int)u8 == strides[l]) { lvl = l; break; }
    }
    level[i] = lvl;
    if (rois_per_level && lvl >= 0)
      memcpy(rois_per_level + ((long)lvl * n_rois + i) * 4, r, 4 * sizeof(float));
  }
}

/* models/FPN/builder.py:563-610: per level X.roi_align(feat_l, rois_l, out, stride) then add_n.
 * maxidx outputs of the four ops are merged: the assigned level's planes (others are all -1). */
void orc_fpn_roi_align_fwd(const float* const* feats, const int* Hs, const int* Ws,
                           const int* strides, int nlvl, const float* rois, float* out,
                           float* amax_x, float* amax_y, int B, int C, int R, int ph, int pw,
                           float canonical_scale, float canonical_level, int nthreads) {
  int n_rois = B * R;
  size_t per_roi = (size_t)C * ph * pw, total = per_roi * n_rois;
  int* level = (int*)malloc(sizeof(int) * n_rois);
  float* rl = (float*)malloc(sizeof(float) * (size_t)nlvl * n_rois * 4);
  float* t_o = (float*)malloc(sizeof(float) * total);
  float* t_x = (float*)malloc(sizeof(float) * total);
  float* t_y = (float*)malloc(sizeof(float) * total);
  orc_fpn_roi_assign(rois, n_rois, strides, nlvl, canonical_scale, canonical_level, level, rl);
  /* the glue loops (fill, add_n, argmax merge) run on the same threads as the operator itself, so
   * that the multi-thread cpu_baseline is not dominated by serial memory passes */
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
  for (long i = 0; i < (long)total; ++i) { amax_x[i] = -1.f; amax_y[i] = -1.f; }
  for (int l = 0; l < nlvl; ++l) {
    orc_roi_align_v2_fwd(feats[l], rl + (size_t)l * n_rois * 4, t_o, t_x, t_y, B, C, Hs[l], Ws[l],
                         R, ph, pw, 1.0f / (float)strides[l], nthreads);
    /* add_n = ElementWiseSum in input order */
    if (l == 0) {
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
      for (long i = 0; i < (long)total; ++i) out[i] = t_o[i];
    } else {
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
      for (long i = 0; i < (long)total; ++i) out[i] = out[i] + t_o[i];
    }
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
    for (int r = 0; r < n_rois; ++r)
      if (level[r] == l) {
        memcpy(amax_x + r * per_roi, t_x + r * per_roi, sizeof(float) * per_roi);
        memcpy(amax_y + r * per_roi, t_y + r * per_roi, sizeof(float) * per_roi);
      }
  }
  free(level); free(rl); free(t_o); free(t_x); free(t_y);
}

void orc_fpn_roi_align_bwd(const float* dy, const float* rois, const float* amax_x,
                           const float* amax_y, float* const* dfeats, const int* Hs,
                           const int* Ws, const int* strides, int nlvl, int B, int C, int R,
                           int ph, int pw, float canonical_scale, float canonical_level, int req,
                           int nthreads) {
  int n_rois = B * R;
  size_t per_roi = (size_t)C * ph * pw, total = per_roi * n_rois;
  int* level = (int*)malloc(sizeof(int) * n_rois);
  float* t_x = (float*)malloc(sizeof(float) * total);
  float* t_y = (float*)malloc(sizeof(float) * total);
  orc_fpn_roi_assign(rois, n_rois, strides, nlvl, canonical_scale, canonical_level, level, NULL);
  for (int l = 0; l < nlvl; ++l) {
    /* the level-l op only holds argmax for its own RoIs; everything else is -1 */
#pragma omp parallel for num_threads(nthreads > 0 ? nthreads : 1) schedule(static)
    for (int r = 0; r < n_rois; ++r)
      for (size_t i = 0; i < per_roi; ++i) {
        t_x[r * per_roi + i] = level[r] == l ? amax_x[r * per_roi + i] : -1.f;
        t_y[r * per_roi + i] = level[r] == l ? amax_y[r * per_roi + i] : -1.f;
      }
    orc_roi_align_v2_bwd(dy, t_x, t_y, dfeats[l], B, C, Hs[l], Ws[l], R, ph, pw, req, nthreads);
  }
  free(level); free(t_x); free(t_y);
}
