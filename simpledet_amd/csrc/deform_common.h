// DeformableConvolution (v1) for gfx950: what the translation units share.
//   deform_sample.hip   deformable im2col / col2im / col2im_coord (the sampling kernels) + their entry points
//   deform_gemm.hip     fp32-in / fp32-out matrix-core GEMM (fp32 MFMA and the scaled fp16 / bf16 splits), operand maxima
//   deform_fused.hip    the forward without a col matrix (sampling fused into the GEMM's B-operand staging)
//   deform_conv.hip     the operator: forward / backward over those pieces, num_group, bias
//   deform_split.h      the hi / lo split arithmetic the GEMM and the fused forward share
#pragma once
#include "common.h"
#include "../../include/simpledet_ops.h"
#include <math.h>
#include <float.h>
#include <type_traits>

namespace sd {

struct DcnGeom {
  int N, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, dgroup, Ho, Wo;
};

// sampling position of (tap, pixel) and the in-bounds test of deformable_im2col_gpu_kernel
struct Sample {
  bool ok;
  int h_low, w_low, h_high, w_high;
  float w1, w2, w3, w4;
  float lh, lw;   // the fractions the four weights are products of: w1 = (1 - lh)(1 - lw), w2 = (1 - lh) lw, w3 = lh (1 - lw), w4 = lh lw
};

__device__ __forceinline__ Sample im2col_sample(const DcnGeom& g, int h_in, int w_in, int i, int j,
                                                float offset_h, float offset_w) {
  Sample s;
  const float h_im = h_in + i * g.dil_h + offset_h;
  const float w_im = w_in + j * g.dil_w + offset_w;
  s.ok = h_im >= 0 && w_im >= 0 && h_im < g.H && w_im < g.W;
  // deformable_im2col_bilinear on the patch-relative coordinates (map_h, map_w)
  float h = i * g.dil_h + offset_h, w = j * g.dil_w + offset_w;
  const int height = g.H - h_in, width = g.W - w_in;
  int h_low = (int)floorf(h), w_low = (int)floorf(w), h_high, w_high;
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
  const float lh = h - h_low, lw = w - w_low, hh = 1 - lh, hw = 1 - lw;
  s.w1 = hh * hw; s.w2 = hh * lw; s.w3 = lh * hw; s.w4 = lh * lw;
  s.lh = lh; s.lw = lw;
  s.h_low = h_low + h_in; s.h_high = h_high + h_in;  // absolute rows / columns
  s.w_low = w_low + w_in; s.w_high = w_high + w_in;
  return s;
}

constexpr int kDcnMaxTaps = 9;        // taps whose sampling state the LDS-plane kernels keep in registers

// packed corner state: bits 0-27 index of (h_low, w_low), bit 28 w_high - w_low, bit 29
// h_high - h_low, bit 30 "inside the image"; 0 = outside (reads corner 0, contributes exactly 0)
constexpr int kDcnInside = 1 << 30;
__device__ __forceinline__ int dcn_pack(bool ok, int h_low, int w_low, int h_high, int w_high,
                                        int W) {
  if (!ok) return 0;
  return (h_low * W + w_low) | ((w_high - w_low) << 28) | ((h_high - h_low) << 29) | kDcnInside;
}

// max|x| of a (batch, rows, cols) operand with row stride ld and batch stride bstride (deform_gemm.hip)
struct AbsSeg {
  const float* p;
  long rows;
  int cols;
  long ld, bstride;
  int batch;
  unsigned* out;
  int blocks;   // workgroups of the launch that work on this operand
};
AbsSeg absmax_seg(const float* p, long rows, int cols, long ld, long bstride, int batch, unsigned* out);
void launch_absmax(AbsSeg s0, AbsSeg s1 = AbsSeg{}, AbsSeg s2 = AbsSeg{}, hipStream_t st = nullptr);

// ---- host functions used across the translation units ----
int make_geom(DcnGeom& g, int N, int C, int H, int W, int kh, int kw, int pad_h, int pad_w, int stride_h,
              int stride_w, int dil_h, int dil_w, int dgroup);
// wsum (device, N * dgroup words, or null): with it the four-channel col2im sums in fixed point
int col2im_impl(const float* col, const float* offset, float* dx, int req, int N, int C, int H, int W, int kh,
                int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w, int dgroup,
                void* stream, unsigned* wsum);
int gemm_f32_impl(int transA, int transB, int M, int N, int K, const float* A, int lda, long strideA,
                  const float* B, int ldb, long strideB, float* C, int ldc, long strideC, int batch,
                  int accumulate, const unsigned* amax, void* stream);
// deform_fused.hip: the col-free forward (falls back to `unfused` for shapes it does not take)
bool dcn_fused_shape_ok(int C, int H, int W, int kh, int kw, int dgroup);
int deform_conv_fwd_nocol_impl(const float* x, const float* offset, const float* weight, const float* bias,
                               float* y, int N, int C, int H, int W, int F, int kh, int kw, int pad, int stride,
                               int dil, int dgroup, void* workspace, size_t workspace_bytes, void* stream);
// deform_conv.hip: im2col + one GEMM per group (+ bias pass); the col matrix stays in the workspace
int deform_conv_fwd_impl(const float* x, const float* offset, const float* weight, const float* bias, float* y,
                         int N, int C, int H, int W, int F, int kh, int kw, int pad, int stride, int dil,
                         int dgroup, int num_group, void* workspace, size_t workspace_bytes, void* stream);

}  // namespace sd
