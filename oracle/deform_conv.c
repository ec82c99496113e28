/*
 * ORACLE (test infrastructure, not product code) -- CPU restatement of DeformableConvolution (v1).
 *
 * PARITY UNPINNED: the arithmetic lives in a third-party dependency that is not vendored under
 * /root/reference: apache/incubator-mxnet pinned at branch 1.6.0 (docker/Dockerfile:48; wheels
 * mxnet_cu101-1.6.0b20191214, README.md:58), operator src/operator/contrib/deformable_convolution
 * {-inl.h,.cu} + src/operator/contrib/nn/deformable_im2col.cuh.  This file restates the PUBLISHED
 * algorithm of that version (Dai et al., "Deformable Convolutional Networks", ICCV 2017, as released
 * in msracver/Deformable-ConvNets and merged into MXNet):
 *   deformable_im2col_gpu_kernel      -> orc_deform_im2col
 *   deformable_col2im_gpu_kernel      -> orc_deform_col2im        (data gradient)
 *   deformable_col2im_coord_gpu_kernel-> orc_deform_col2im_coord  (offset gradient)
 *   forward = W(F x C*kh*kw) . col(C*kh*kw x Ho*Wo), group = 1, no bias      -> orc_deform_conv_fwd
 *   DeformableConvolutionOp::Forward / ::Backward with every parameter (num_group: one product per
 *   filter block over its channel block's col rows; bias: `out += broadcast<1>(bias)`;
 *   gbias = sumall_except_dim<1>(grad))                    -> orc_deform_convolution_fwd / _bwd
 *   (the bias / num_group call sites: models/RepPoints/builder.py:215-245, models/sepc/sepc_dconv.py:5-16)
 * The reference's call sites fix the configuration: models/dcn/builder.py:14-17 (3x3, pad = dilate,
 * num_deformable_group = 4, no_bias, fp32).  There is no golden vector for it anywhere in the
 * reference; tests additionally check properties that do not depend on this restatement (zero
 * offsets == ordinary convolution, integer offsets == shifted convolution, adjointness of
 * im2col/col2im, finite differences for the offset gradient).
 *
 * Layouts: x (C,H,W); offset (dgroup*2*kh*kw, Ho, Wo) with channel 2*(i*kw+j) = dh, +1 = dw;
 * col (C*kh*kw, Ho*Wo) with row (c*kh + i)*kw + j.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* deformable_im2col_bilinear: sample relative to the patch origin, clamped at the far border */
static float im2col_bilinear(const float* bottom_data, int data_width, int height, int width,
                             float h, float w) {
  int h_low = (int)floorf(h);
  int w_low = (int)floorf(w);
  int h_high, w_high;
  if (h_low >= height - 1) {
    h_high = h_low = height - 1;
    h = (float)h_low;
  } else {
    h_high = h_low + 1;
  }
  if (w_low >= width - 1) {
    w_high = w_low = width - 1;
    w = (float)w_low;
  } else {
    w_high = w_low + 1;
  }
  float lh = h - h_low;
  float lw = w - w_low;
  float hh = 1 - lh, hw = 1 - lw;
  float v1 = bottom_data[h_low * data_width + w_low];
  float v2 = bottom_data[h_low * data_width + w_high];
  float v3 = bottom_data[h_high * data_width + w_low];
  float v4 = bottom_data[h_high * data_width + w_high];
  float w1 = hh * hw, w2 = hh * lw, w3 = lh * hw, w4 = lh * lw;
  float val = (w1 * v1 + w2 * v2 + w3 * v3 + w4 * v4);
  return val;
}

void orc_deform_im2col(const float* x, const float* offset, float* col, int C, int H, int W,
                       int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w,
                       int dil_h, int dil_w, int dgroup, int Ho, int Wo) {
  const int cpg = C / dgroup; /* channel_per_deformable_group */
  for (int c_im = 0; c_im < C; ++c_im)
    for (int h_col = 0; h_col < Ho; ++h_col)
      for (int w_col = 0; w_col < Wo; ++w_col) {
        const int c_col = c_im * kh * kw;
        const int g = c_im / cpg;
        const int h_in = h_col * stride_h - pad_h;
        const int w_in = w_col * stride_w - pad_w;
        float* data_col_ptr = col + ((long)c_col * Ho + h_col) * Wo + w_col;
        const float* data_im_ptr = x + ((long)c_im * H + h_in) * W + w_in;
        const float* data_offset_ptr = offset + (long)g * 2 * kh * kw * Ho * Wo;
        for (int i = 0; i < kh; ++i)
          for (int j = 0; j < kw; ++j) {
            const long oh = ((long)(2 * (i * kw + j)) * Ho + h_col) * Wo + w_col;
            const long ow = ((long)(2 * (i * kw + j) + 1) * Ho + h_col) * Wo + w_col;
            const float offset_h = data_offset_ptr[oh];
            const float offset_w = data_offset_ptr[ow];
            float val = 0.f;
            const float h_im = h_in + i * dil_h + offset_h;
            const float w_im = w_in + j * dil_w + offset_w;
            if (h_im >= 0 && w_im >= 0 && h_im < H && w_im < W) {
              const float map_h = i * dil_h + offset_h;
              const float map_w = j * dil_w + offset_w;
              const int cur_height = H - h_in;
              const int cur_width = W - w_in;
              val = im2col_bilinear(data_im_ptr, W, cur_height, cur_width, map_h, map_w);
            }
            *data_col_ptr = val;
            data_col_ptr += (long)Ho * Wo;
          }
      }
}

static float get_gradient_weight(float argmax_h, float argmax_w, int h, int w, int height,
                                 int width) {
  if (argmax_h < 0 || argmax_h > height || argmax_w < 0 || argmax_w > width) return 0;
  argmax_h = argmax_h > 0.f ? argmax_h : 0.f;
  argmax_w = argmax_w > 0.f ? argmax_w : 0.f;
  int argmax_h_low = (int)argmax_h;
  int argmax_w_low = (int)argmax_w;
  int argmax_h_high, argmax_w_high;
  if (argmax_h_low >= height - 1) {
    argmax_h_high = argmax_h_low = height - 1;
    argmax_h = (float)argmax_h_low;
  } else {
    argmax_h_high = argmax_h_low + 1;
  }
  if (argmax_w_low >= width - 1) {
    argmax_w_high = argmax_w_low = width - 1;
    argmax_w = (float)argmax_w_low;
  } else {
    argmax_w_high = argmax_w_low + 1;
  }
  float weight = 0;
  if (h == argmax_h_low) {
    if (w == argmax_w_low) weight = (h + 1 - argmax_h) * (w + 1 - argmax_w);
    else if (w == argmax_w_high) weight = (h + 1 - argmax_h) * (argmax_w + 1 - w);
  } else if (h == argmax_h_high) {
    if (w == argmax_w_low) weight = (argmax_h + 1 - h) * (w + 1 - argmax_w);
    else if (w == argmax_w_high) weight = (argmax_h + 1 - h) * (argmax_w + 1 - w);
  }
  return weight;
}

/* dx += scatter of col (the caller zero-fills for kWriteTo) */
void orc_deform_col2im(const float* col, const float* offset, float* dx, int C, int H, int W,
                       int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w,
                       int dil_h, int dil_w, int dgroup, int Ho, int Wo) {
  const int cpg = C / dgroup;
  const long n = (long)C * kh * kw * Ho * Wo;
  for (long index = 0; index < n; ++index) {
    const int j = (int)((index / Wo / Ho) % kw);
    const int i = (int)((index / Wo / Ho / kw) % kh);
    const int c = (int)(index / Wo / Ho / kw / kh);
    const int g = c / cpg;
    const int w_out = (int)(index % Wo);
    const int h_out = (int)((index / Wo) % Ho);
    const int w_in = w_out * stride_w - pad_w;
    const int h_in = h_out * stride_h - pad_h;
    const float* data_offset_ptr = offset + (long)g * 2 * kh * kw * Ho * Wo;
    const float offset_h = data_offset_ptr[((long)(2 * (i * kw + j)) * Ho + h_out) * Wo + w_out];
    const float offset_w = data_offset_ptr[((long)(2 * (i * kw + j) + 1) * Ho + h_out) * Wo + w_out];
    const float cur_inv_h_data = h_in + i * dil_h + offset_h;
    const float cur_inv_w_data = w_in + j * dil_w + offset_w;
    const float cur_top_grad = col[index];
    const int cur_h = (int)cur_inv_h_data;
    const int cur_w = (int)cur_inv_w_data;
    for (int dy = -2; dy <= 2; dy++)
      for (int dxx = -2; dxx <= 2; dxx++) {
        if (cur_h + dy >= 0 && cur_h + dy < H && cur_w + dxx >= 0 && cur_w + dxx < W &&
            fabsf(cur_inv_h_data - (cur_h + dy)) < 1 && fabsf(cur_inv_w_data - (cur_w + dxx)) < 1) {
          long pos = ((long)c * H + cur_h + dy) * W + cur_w + dxx;
          float weight = get_gradient_weight(cur_inv_h_data, cur_inv_w_data, cur_h + dy,
                                             cur_w + dxx, H, W);
          dx[pos] += weight * cur_top_grad;
        }
      }
  }
}

static float get_coordinate_weight(float argmax_h, float argmax_w, int height, int width,
                                   const float* im_data, int data_width, int bp_dir) {
  if (argmax_h < 0 || argmax_h > height || argmax_w < 0 || argmax_w > width) return 0;
  if (argmax_h < 0) argmax_h = 0;
  if (argmax_w < 0) argmax_w = 0;
  int argmax_h_low = (int)argmax_h;
  int argmax_w_low = (int)argmax_w;
  int argmax_h_high, argmax_w_high;
  if (argmax_h_low >= height - 1) {
    argmax_h_high = argmax_h_low = height - 1;
    argmax_h = (float)argmax_h_low;
  } else {
    argmax_h_high = argmax_h_low + 1;
  }
  if (argmax_w_low >= width - 1) {
    argmax_w_high = argmax_w_low = width - 1;
    argmax_w = (float)argmax_w_low;
  } else {
    argmax_w_high = argmax_w_low + 1;
  }
  float weight = 0;
  if (bp_dir == 0) {
    weight += -1 * (argmax_w_low + 1 - argmax_w) * im_data[argmax_h_low * data_width + argmax_w_low];
    weight += -1 * (argmax_w - argmax_w_low) * im_data[argmax_h_low * data_width + argmax_w_high];
    weight += (argmax_w_low + 1 - argmax_w) * im_data[argmax_h_high * data_width + argmax_w_low];
    weight += (argmax_w - argmax_w_low) * im_data[argmax_h_high * data_width + argmax_w_high];
  } else if (bp_dir == 1) {
    weight += -1 * (argmax_h_low + 1 - argmax_h) * im_data[argmax_h_low * data_width + argmax_w_low];
    weight += (argmax_h_low + 1 - argmax_h) * im_data[argmax_h_low * data_width + argmax_w_high];
    weight += -1 * (argmax_h - argmax_h_low) * im_data[argmax_h_high * data_width + argmax_w_low];
    weight += (argmax_h - argmax_h_low) * im_data[argmax_h_high * data_width + argmax_w_high];
  }
  return weight;
}

void orc_deform_col2im_coord(const float* col, const float* x, const float* offset, float* doff,
                             int C, int H, int W, int kh, int kw, int pad_h, int pad_w,
                             int stride_h, int stride_w, int dil_h, int dil_w, int dgroup, int Ho,
                             int Wo) {
  const int cpg_col = C * kh * kw / dgroup; /* col rows per deformable group */
  const long n = (long)Ho * Wo * 2 * kh * kw * dgroup;
  for (long index = 0; index < n; ++index) {
    float val = 0;
    const int w = (int)(index % Wo);
    const int h = (int)((index / Wo) % Ho);
    const int c = (int)(index / Wo / Ho);
    const int g = c / (2 * kh * kw);
    const int col_step = kh * kw;
    int cnt = 0;
    const float* data_col_ptr = col + (long)g * cpg_col * Wo * Ho;
    const float* data_im_ptr = x + (long)g * (cpg_col / kh / kw) * H * W;
    const float* data_offset_ptr = offset + (long)g * 2 * kh * kw * Ho * Wo;
    const int offset_c = c - g * 2 * kh * kw;
    for (int col_c = (offset_c / 2); col_c < cpg_col; col_c += col_step) {
      const long col_pos = (((long)col_c * Ho) + h) * Wo + w;
      const int bp_dir = offset_c % 2;
      int j = (int)((col_pos / Wo / Ho) % kw);
      int i = (int)((col_pos / Wo / Ho / kw) % kh);
      int w_out = (int)(col_pos % Wo);
      int h_out = (int)((col_pos / Wo) % Ho);
      int w_in = w_out * stride_w - pad_w;
      int h_in = h_out * stride_h - pad_h;
      const float offset_h = data_offset_ptr[((long)(2 * (i * kw + j)) * Ho + h_out) * Wo + w_out];
      const float offset_w = data_offset_ptr[((long)(2 * (i * kw + j) + 1) * Ho + h_out) * Wo + w_out];
      float inv_h = h_in + i * dil_h + offset_h;
      float inv_w = w_in + j * dil_w + offset_w;
      if (inv_h < 0 || inv_w < 0 || inv_h >= H || inv_w >= W) inv_h = inv_w = -1;
      const float weight = get_coordinate_weight(inv_h, inv_w, H, W,
                                                 data_im_ptr + (long)cnt * H * W, W, bp_dir);
      val += weight * data_col_ptr[col_pos];
      cnt += 1;
    }
    doff[index] = val;
  }
}

void orc_deform_conv_fwd(const float* x, const float* offset, const float* wt, float* y, int N,
                         int C, int H, int W, int F, int kh, int kw, int pad, int stride, int dil,
                         int dgroup) {
  const int Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;
  const int Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
  const long K = (long)C * kh * kw, P = (long)Ho * Wo;
  float* col = (float*)malloc(sizeof(float) * (size_t)(K * P));
  for (int n = 0; n < N; ++n) {
    orc_deform_im2col(x + (long)n * C * H * W, offset + (long)n * dgroup * 2 * kh * kw * P, col, C,
                      H, W, kh, kw, pad, pad, stride, stride, dil, dil, dgroup, Ho, Wo);
    float* yn = y + (long)n * F * P;
#pragma omp parallel for
    for (long f = 0; f < F; ++f) {
      float* yr = yn + f * P;
      memset(yr, 0, sizeof(float) * (size_t)P);
      for (long k = 0; k < K; ++k) {
        const float a = wt[f * K + k];
        const float* cr = col + k * P;
        for (long p = 0; p < P; ++p) yr[p] += a * cr[p];
      }
    }
  }
  free(col);
}

/* DeformableConvolutionOp::Forward (deformable_convolution-inl.h, upstream 1.6.0): per image im2col, per
 * group g  output_3d[g] = dot(weight_3d[g], col_buffer_3d[g]);  then, with a bias,
 * out += broadcast<1>(bias).  weight (F, C/G, kh, kw); bias NULL = no_bias.  fp32 sums in k order. */
void orc_deform_convolution_fwd(const float* x, const float* offset, const float* wt, const float* bias,
                                float* y, int N, int C, int H, int W, int F, int kh, int kw, int pad,
                                int stride, int dil, int dgroup, int num_group) {
  const int Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;
  const int Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
  const long K = (long)C * kh * kw, P = (long)Ho * Wo, Kg = K / num_group, Fg = F / num_group;
  float* col = (float*)malloc(sizeof(float) * (size_t)(K * P));
  for (int n = 0; n < N; ++n) {
    orc_deform_im2col(x + (long)n * C * H * W, offset + (long)n * dgroup * 2 * kh * kw * P, col, C,
                      H, W, kh, kw, pad, pad, stride, stride, dil, dil, dgroup, Ho, Wo);
    float* yn = y + (long)n * F * P;
#pragma omp parallel for
    for (long f = 0; f < F; ++f) {
      const long g = f / Fg;
      float* yr = yn + f * P;
      memset(yr, 0, sizeof(float) * (size_t)P);
      for (long k = 0; k < Kg; ++k) {
        const float a = wt[f * Kg + k];
        const float* cr = col + (g * Kg + k) * P;
        for (long p = 0; p < P; ++p) yr[p] += a * cr[p];
      }
      if (bias)
        for (long p = 0; p < P; ++p) yr[p] += bias[f];
    }
  }
  free(col);
}

/* DeformableConvolutionOp::Backward: per image  col_buffer_3d[g] = dot(weight_3d[g].T, out_grad_3d[g]),
 * deformable_col2im_coord -> d_offset, deformable_col2im -> d_x, then im2col again and
 * dweight_3d[g] (+)= dot(out_grad_3d[g], col_buffer_3d[g].T) image after image; with a bias
 * gbias = sumall_except_dim<1>(grad).  All four gradients are written (req = write); dcol in fp32
 * with the sums in f order, dW and gbias in double (the reference sums them in fp32 in an order the
 * library does not fix: the tests hold the product to a tolerance, not to bits). */
void orc_deform_convolution_bwd(const float* dy, const float* x, const float* offset, const float* wt,
                                float* dx, float* doff, float* dw, float* dbias, int N, int C, int H, int W,
                                int F, int kh, int kw, int pad, int stride, int dil, int dgroup,
                                int num_group) {
  const int Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;
  const int Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
  const long K = (long)C * kh * kw, P = (long)Ho * Wo, Kg = K / num_group, Fg = F / num_group;
  float* col = (float*)malloc(sizeof(float) * (size_t)(K * P));
  double* dwd = (double*)calloc((size_t)(F * Kg), sizeof(double));
  for (int n = 0; n < N; ++n) {
    const float* xn = x + (long)n * C * H * W;
    const float* on = offset + (long)n * dgroup * 2 * kh * kw * P;
    const float* dyn = dy + (long)n * F * P;
#pragma omp parallel for
    for (long k = 0; k < K; ++k) {
      const long g = k / Kg, kk = k - g * Kg;
      float* cr = col + k * P;
      memset(cr, 0, sizeof(float) * (size_t)P);
      for (long f = g * Fg; f < (g + 1) * Fg; ++f) {
        const float a = wt[f * Kg + kk];
        const float* dr = dyn + f * P;
        for (long p = 0; p < P; ++p) cr[p] += a * dr[p];
      }
    }
    orc_deform_col2im_coord(col, xn, on, doff + (long)n * dgroup * 2 * kh * kw * P, C, H, W, kh, kw, pad, pad,
                            stride, stride, dil, dil, dgroup, Ho, Wo);
    memset(dx + (long)n * C * H * W, 0, sizeof(float) * (size_t)C * H * W);
    orc_deform_col2im(col, on, dx + (long)n * C * H * W, C, H, W, kh, kw, pad, pad, stride, stride, dil, dil,
                      dgroup, Ho, Wo);
    orc_deform_im2col(xn, on, col, C, H, W, kh, kw, pad, pad, stride, stride, dil, dil, dgroup, Ho, Wo);
#pragma omp parallel for
    for (long f = 0; f < F; ++f) {
      const long g = f / Fg;
      const float* dr = dyn + f * P;
      for (long k = 0; k < Kg; ++k) {
        const float* cr = col + (g * Kg + k) * P;
        double acc = 0.0;
        for (long p = 0; p < P; ++p) acc += (double)dr[p] * (double)cr[p];
        dwd[f * Kg + k] += acc;
      }
    }
  }
  for (long i = 0; i < F * Kg; ++i) dw[i] = (float)dwd[i];
  if (dbias)
    for (long f = 0; f < F; ++f) {
      double acc = 0.0;
      for (int n = 0; n < N; ++n) {
        const float* dr = dy + ((long)n * F + f) * P;
        for (long p = 0; p < P; ++p) acc += dr[p];
      }
      dbias[f] = (float)acc;
    }
  free(dwd);
  free(col);
}
