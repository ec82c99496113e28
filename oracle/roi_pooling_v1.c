/*
 * ORACLE (test infrastructure, not product code) -- CPU restatement of ROIPooling_v1.
 *
 * Follows (DType = float):
 *   forward   operator_cxx/roi_pooling_v1.cc:39-126 (ROIPoolForward_v1) with the op wrapper's
 *             pre-fill out = -FLT_MAX, max_idx = -1 (roi_pooling_v1-inl.h:91-92)
 *   backward  operator_cxx/roi_pooling_v1.cu:115-152 (ROIPoolBackward, scatter) -- the GPU kernel
 *             is the spec of the device op; launcher zero-fill/add roi_pooling_v1-inl.h:124-129
 *   backward' operator_cxx/roi_pooling_v1.cc:128-221 (ROIPoolBackwardAcc_v1, CPU gather form)
 * Pinned by the reference's own docstring example (roi_pooling_v1.cc:265-285), see
 * tests/test_roi_pool.py::test_reference_docstring_golden.
 */
#include "oracle.h"
#include <float.h>
#include <math.h>
#include <string.h>

static inline int imax(int a, int b) { return a > b ? a : b; }
static inline int imin(int a, int b) { return a < b ? a : b; }

void orc_roi_pool_v1_fwd(const float* data, const float* rois, float* out, float* maxidx, int B,
                         int C, int H, int W, int K, int ph_, int pw_, float spatial_scale) {
  (void)B;
  const long data_size = (long)C * H * W;
  /* roi_pooling_v1-inl.h:91-92 */
  for (long i = 0; i < (long)K * C * ph_ * pw_; ++i) {
    out[i] = -FLT_MAX;
    maxidx[i] = -1.0f;
  }
  const float* bottom_rois = rois;
  float* top_data = out;
  float* argmax_data = maxidx;
  for (int n = 0; n < K; ++n) { /* roi_pooling_v1.cc:58 */
    int roi_batch_ind = (int)bottom_rois[0];
    int roi_start_w = (int)round(bottom_rois[1] * spatial_scale); /* double round(), :60-63 */
    int roi_start_h = (int)round(bottom_rois[2] * spatial_scale);
    int roi_end_w = (int)round(bottom_rois[3] * spatial_scale);
    int roi_end_h = (int)round(bottom_rois[4] * spatial_scale);
    int roi_height = imax(roi_end_h - roi_start_h + 1, 1); /* :68-69 */
    int roi_width = imax(roi_end_w - roi_start_w + 1, 1);
    const float bin_size_h = (float)roi_height / (float)ph_;
    const float bin_size_w = (float)roi_width / (float)pw_;
    const float* batch_data = data + data_size * roi_batch_ind;
    for (int c = 0; c < C; ++c) {
      for (int ph = 0; ph < ph_; ++ph) {
        for (int pw = 0; pw < pw_; ++pw) {
          int hstart = (int)floorf((float)ph * bin_size_h); /* :83-90, std::floor(float) */
          int wstart = (int)floorf((float)pw * bin_size_w);
          int hend = (int)ceilf((float)(ph + 1) * bin_size_h);
          int wend = (int)ceilf((float)(pw + 1) * bin_size_w);
          hstart = imin(imax(hstart + roi_start_h, 0), H); /* :92-95 */
          hend = imin(imax(hend + roi_start_h, 0), H);
          wstart = imin(imax(wstart + roi_start_w, 0), W);
          wend = imin(imax(wend + roi_start_w, 0), W);
          int is_empty = (hend <= hstart) || (wend <= wstart);
          const int pool_index = ph * pw_ + pw;
          if (is_empty) { /* :100-103 */
            top_data[pool_index] = 0;
            argmax_data[pool_index] = -1;
          }
          for (int h = hstart; h < hend; ++h)
            for (int w = wstart; w < wend; ++w) {
              const int index = h * W + w;
              if (batch_data[index] > top_data[pool_index]) { /* :108-111 */
                top_data[pool_index] = batch_data[index];
                argmax_data[pool_index] = (float)index;
              }
            }
        }
      }
      batch_data += (long)H * W;
      top_data += ph_ * pw_;
      argmax_data += ph_ * pw_;
    }
    bottom_rois += 5;
  }
}

/* roi_pooling_v1.cu:115-152: dX[roi_batch, c, argmax] += dY  (atomicAdd; order-free for the
 * oracle because each addend is exact in the fp32 sum only up to rounding: tests allow 1e-4) */
void orc_roi_pool_v1_bwd(const float* dy, const float* rois, const float* maxidx, float* dx,
                         int B, int C, int H, int W, int K, int ph_, int pw_, float spatial_scale,
                         int req) {
  (void)spatial_scale;
  if (req == 1) memset(dx, 0, sizeof(float) * (size_t)B * C * H * W);
  const int PP = ph_ * pw_;
  for (long index = 0; index < (long)K * C * PP; ++index) {
    int c = (int)((index / PP) % C);
    int n = (int)(index / PP / C);
    int roi_batch_ind = (int)rois[(long)n * 5];
    int argmax = (int)maxidx[index];
    if (argmax != -1) dx[((long)roi_batch_ind * C + c) * H * W + argmax] += dy[index];
  }
}

/* roi_pooling_v1.cc:128-221 */
void orc_roi_pool_v1_bwd_cpu_gather(const float* dy, const float* rois, const float* maxidx,
                                    float* dx, int B, int C, int H, int W, int K, int ph_, int pw_,
                                    float spatial_scale, int req) {
  if (req == 1) memset(dx, 0, sizeof(float) * (size_t)B * C * H * W);
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c)
      for (int h = 0; h < H; ++h)
        for (int w = 0; w < W; ++w) {
          long off = ((long)(b * C + c) * H + h) * W + w;
          float gradient = 0;
          for (int roi_n = 0; roi_n < K; ++roi_n) {
            const float* r = rois + (long)roi_n * 5;
            if (b != (int)r[0]) continue;
            int roi_start_w = (int)round(r[1] * spatial_scale);
            int roi_start_h = (int)round(r[2] * spatial_scale);
            int roi_end_w = (int)round(r[3] * spatial_scale);
            int roi_end_h = (int)round(r[4] * spatial_scale);
            int in_roi = (w >= roi_start_w && w <= roi_end_w && h >= roi_start_h && h <= roi_end_h);
            if (!in_roi) continue;
            int roi_height = imax(roi_end_h - roi_start_h + 1, 1);
            int roi_width = imax(roi_end_w - roi_start_w + 1, 1);
            const float bin_size_h = (float)roi_height / (float)ph_;
            const float bin_size_w = (float)roi_width / (float)pw_;
            int phstart = (int)floorf((float)(h - roi_start_h) / bin_size_h);
            int pwstart = (int)floorf((float)(w - roi_start_w) / bin_size_w);
            int phend = (int)ceilf((float)(h - roi_start_h + 1) / bin_size_h);
            int pwend = (int)ceilf((float)(w - roi_start_w + 1) / bin_size_w);
            phstart = imin(imax(phstart, 0), ph_);
            phend = imin(imax(phend, 0), ph_);
            pwstart = imin(imax(pwstart, 0), pw_);
            pwend = imin(imax(pwend, 0), pw_);
            long offset = ((long)roi_n * C + c) * ph_ * pw_;
            for (int ph = phstart; ph < phend; ++ph)
              for (int pw = pwstart; pw < pwend; ++pw) {
                const int pooled_index = ph * pw_ + pw;
                if ((int)maxidx[offset + pooled_index] == h * W + w)
                  gradient += dy[offset + pooled_index];
              }
          }
          dx[off] += gradient;
        }
}
