// DeformableConvolution (v1) for gfx950: deformable im2col / col2im / col2im_coord + fp32 MFMA GEMM.
//   reference call site: models/dcn/builder.py:14-17 (3x3, pad = dilate, num_deformable_group 4,
//   no_bias, fp32).  The arithmetic is upstream MXNet 1.6.0 (src/operator/contrib/nn/
//   deformable_im2col.cuh, deformable_convolution-inl.h), NOT vendored in the reference tree:
//   parity is against the restated published algorithm (oracle/deform_conv.c, "parity unpinned").
// MI355X design
//   * im2col: the sampling position and the four bilinear weights of (pixel, tap) are shared by
//     all C/dgroup channels of a deformable group, so one lane owns (pixel, tap, group), computes
//     them once and streams the channels: 4 gathers + 7 flops per element, stores contiguous along
//     the pixel axis (the col matrix is the HBM-bound stream: 4*9*C*Ho*Wo bytes per image).
//   * col2im / col2im_coord: same ownership; the data gradient is scattered with hardware fp32
//     atomics (as the reference does), the offset gradient is a per-lane reduction over channels.
//   * GEMM: the only MFMA work on the hot path, fp32 in / fp32 out (the reference computes this
//     layer in fp32: models/tridentnet/resnet_v1.py:209).  Default: every fp32 product as three bf16
//     MFMA terms of a hi/lo operand split (gemm_f32_split_bf16_kernel below: 4.5e-6 x max|C| on the
//     DCN products, 2.5-2.9x the fp32 MFMA kernel); `deform_gemm_split = 0`: exact fp32 products on
//     v_mfma_f32_32x32x2_f32 (gemm_f32_mfma_kernel: 128 x 64J x 16 tiles, k-major LDS tiles read
//     as conflict-free ds_read_b32).  Both: 4 waves x (2x2) 32x32 accumulators, register-prefetched
//     global loads, XCD-aware tile order that keeps one image's col panel in one L2.
//   (round 5: the sampling kernels, the GEMM and the fused forward live in deform_sample.hip / deform_gemm.hip /
//   deform_fused.hip; this file is the operator over them -- forward / backward, num_group, bias.)
#include "deform_common.h"

namespace sd {

// ---- bias (no_bias = false: models/RepPoints/builder.py:215-245, models/sepc/sepc_dconv.py:12-16) ----
// y[n, f, :] += bias[f], after the products (the order of `out += broadcast<1>(bias)`,
// deformable_convolution-inl.h Forward).  One workgroup per (n, f) row; the fused forward adds the bias
// in its own epilogue and never comes here.
__global__ __launch_bounds__(256) void dcn_bias_add_kernel(float* __restrict__ y, const float* __restrict__ bias,
                                                           int F, int P) {
  const long row = blockIdx.x;
  const float b = bias[row % F];
  float* yr = y + row * P;
  const int head = (int)((4 - (((uintptr_t)yr >> 2) & 3)) & 3);   // floats up to the first 16-byte boundary
  for (int p = threadIdx.x; p < (head < P ? head : P); p += 256) yr[p] += b;
  const int n4 = P > head ? (P - head) >> 2 : 0;
  float4* y4 = reinterpret_cast<float4*>(yr + head);
  for (int i = threadIdx.x; i < n4; i += 256) {
    float4 v = y4[i];
    v.x += b; v.y += b; v.z += b; v.w += b;
    y4[i] = v;
  }
  for (int p = head + 4 * n4 + threadIdx.x; p < P; p += 256) yr[p] += b;
}

// d_bias[f] (+)= sum over n, pixels of out_grad[n, f, :]  (`sumall_except_dim<1>`): one workgroup per
// filter; every lane sums its pixels of every image in a fixed order, then a fixed tree -- the same bits
// in every run.
__global__ __launch_bounds__(256) void dcn_bias_grad_kernel(const float* __restrict__ dy, float* __restrict__ dbias,
                                                            int N, int F, int P, int add) {
  const int f = blockIdx.x;
  float acc = 0.f;
  for (int n = 0; n < N; ++n) {
    const float* r = dy + ((long)n * F + f) * P;
    for (int p = threadIdx.x; p < P; p += 256) acc += r[p];
  }
  __shared__ float sm[256];
  sm[threadIdx.x] = acc;
  __syncthreads();
#pragma unroll
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) dbias[f] = add ? dbias[f] + sm[0] : sm[0];
}

}  // namespace sd

using namespace sd;

extern "C" size_t sd_deform_conv_workspace_bytes(int N, int C, int H, int W, int kh, int kw,
                                                 int pad, int stride, int dil) {
  if (N <= 0 || C <= 0) return 256;
  const long Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;
  const long Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return 256;
  return (size_t)N * C * kh * kw * Ho * Wo * sizeof(float) + 512;
}

// three words behind the (256-byte aligned) col matrix of a DCN workspace: the workspace size
// contract (sd_deform_conv_workspace_bytes) leaves 512 bytes for the alignment and these
static unsigned* dcn_amax_slots(float* col, size_t col_floats) {
  return reinterpret_cast<unsigned*>(((uintptr_t)(col + col_floats) + 15) & ~(uintptr_t)15);
}

static int check_groups(int C, int F, int num_group) {
  SD_REQUIRE(num_group >= 1, "num_group must be positive");
  SD_REQUIRE(C % num_group == 0, "input num_filter must divide group size");
  SD_REQUIRE(F % num_group == 0, "output num_filter must divide group size");
  return SD_OK;
}

static int launch_bias_add(float* y, const float* bias, int N, int F, int P, hipStream_t st) {
  if (!bias || (long)N * F == 0) return SD_OK;
  SD_REQUIRE((long)N * F < (1L << 31), "bias: too many rows");
  hipLaunchKernelGGL(dcn_bias_add_kernel, dim3((unsigned)((long)N * F)), dim3(256), 0, st, y, bias, F, P);
  SD_LAUNCH_CHECK();
  return SD_OK;
}

// forward = im2col + one GEMM per group (+ bias pass); the col matrix stays in the workspace
int sd::deform_conv_fwd_impl(const float* x, const float* offset, const float* weight, const float* bias,
                                float* y, int N, int C, int H, int W, int F, int kh, int kw, int pad, int stride,
                                int dil, int dgroup, int num_group, void* workspace, size_t workspace_bytes,
                                void* stream) {
  DcnGeom g;
  if (int e = make_geom(g, N, C, H, W, kh, kw, pad, pad, stride, stride, dil, dil, dgroup)) return e;
  SD_REQUIRE(F > 0, "num_filter must be positive");
  if (int e = check_groups(C, F, num_group)) return e;
  if (N == 0) return SD_OK;
  SD_REQUIRE(x && offset && weight && y, "null tensor pointer");
  const size_t need = sd_deform_conv_workspace_bytes(N, C, H, W, kh, kw, pad, stride, dil);
  if (!workspace || workspace_bytes < need)
    return fail(SD_ERR_WORKSPACE, "DeformableConvolution workspace too small: %zu < %zu bytes",
                workspace_bytes, need);
  float* col = reinterpret_cast<float*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  if (int e = sd_deform_im2col(x, offset, col, N, C, H, W, kh, kw, pad, pad, stride, stride, dil,
                               dil, dgroup, stream))
    return e;
  const int K = C * kh * kw, P = g.Ho * g.Wo;
  // operand maxima for the scaled fp16 split, behind the col matrix: {max|W|, max|x|}.  A col value is
  // a convex combination of four x values (bilinear weights sum to 1 up to rounding: the split keeps
  // a factor 4 of headroom), so max|x| bounds max|col| without a pass over the 620 MB of col.
  unsigned* amax = dcn_amax_slots(col, (size_t)N * K * P);
  hipStream_t st = (hipStream_t)stream;
  SD_HIP_CHECK(hipMemsetAsync(amax, 0, 16, st));
  launch_absmax(absmax_seg(weight, 1, F * (K / num_group), F * (K / num_group), 0, 1, amax),
                absmax_seg(x, (long)N * C, H * W, H * W, 0, 1, amax + 1), AbsSeg{}, st);
  // y[n][grp] (F/G x P) = W[grp] (F/G x K/G) . col[n][grp] (K/G x P)
  const int Fg = F / num_group, Kg = K / num_group;
  for (int q = 0; q < num_group; ++q)
    if (int e = gemm_f32_impl(0, 0, Fg, P, Kg, weight + (long)q * Fg * Kg, Kg, 0, col + (long)q * Kg * P, P,
                              (long)K * P, y + (long)q * Fg * P, P, (long)F * P, N, 0, amax, stream))
      return e;
  return launch_bias_add(y, bias, N, F, P, st);
}

extern "C" int sd_deform_conv_fwd(const float* x, const float* offset, const float* weight,
                                  float* y, int N, int C, int H, int W, int F, int kh, int kw,
                                  int pad, int stride, int dil, int dgroup, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  return deform_conv_fwd_impl(x, offset, weight, nullptr, y, N, C, H, W, F, kh, kw, pad, stride, dil, dgroup, 1,
                              workspace, workspace_bytes, stream);
}

extern "C" size_t sd_deform_convolution_fwd_workspace_bytes(int N, int C, int H, int W, int F, int kh, int kw,
                                                            int pad, int stride, int dil, int dgroup,
                                                            int num_group, int keep_col) {
  if (!keep_col && num_group == 1)
    return sd_deform_conv_fwd_nocol_workspace_bytes(N, C, H, W, F, kh, kw, pad, stride, dil, dgroup);
  return sd_deform_conv_workspace_bytes(N, C, H, W, kh, kw, pad, stride, dil);
}

extern "C" int sd_deform_convolution_fwd(const float* x, const float* offset, const float* weight,
                                         const float* bias, float* y, int N, int C, int H, int W, int F, int kh,
                                         int kw, int pad, int stride, int dil, int dgroup, int num_group,
                                         int keep_col, void* workspace, size_t workspace_bytes, void* stream) {
  if (!keep_col && num_group == 1)
    return deform_conv_fwd_nocol_impl(x, offset, weight, bias, y, N, C, H, W, F, kh, kw, pad, stride, dil, dgroup,
                                      workspace, workspace_bytes, stream);
  return deform_conv_fwd_impl(x, offset, weight, bias, y, N, C, H, W, F, kh, kw, pad, stride, dil, dgroup,
                              num_group, workspace, workspace_bytes, stream);
}

// `fwd_col`: the col matrix a forward of the same (x, offset) left in ITS workspace
// (sd_deform_conv_col_of_workspace), or null.  With it the backward skips its own im2col
// (0.29 of 1.85 ms on the (16,256,50,84) layer): 620 MB kept per layer between the two calls, which
// 288 GB of HBM make affordable -- the reference recomputes it (deformable_convolution-inl.h
// Backward) because its workspace is shared between operators.
static int deform_conv_bwd_impl(const float* out_grad, const float* x, const float* offset,
                                const float* weight, const float* fwd_col, float* d_x,
                                float* d_offset, float* d_weight, float* d_bias, int req_x, int req_offset,
                                int req_weight, int req_bias, int N, int C, int H, int W, int F, int kh, int kw,
                                int pad, int stride, int dil, int dgroup, int num_group, void* workspace,
                                size_t workspace_bytes, void* stream) {
  DcnGeom g;
  if (int e = make_geom(g, N, C, H, W, kh, kw, pad, pad, stride, stride, dil, dil, dgroup)) return e;
  SD_REQUIRE(F > 0, "num_filter must be positive");
  if (int e = check_groups(C, F, num_group)) return e;
  SD_REQUIRE(req_bias == SD_REQ_NULL || req_bias == SD_REQ_WRITE || req_bias == SD_REQ_ADD, "bad req %d", req_bias);
  if (N == 0) return SD_OK;
  SD_REQUIRE(out_grad && x && offset && weight, "null tensor pointer");
  const size_t need = sd_deform_conv_workspace_bytes(N, C, H, W, kh, kw, pad, stride, dil);
  if (!workspace || workspace_bytes < need)
    return fail(SD_ERR_WORKSPACE, "DeformableConvolution workspace too small: %zu < %zu bytes",
                workspace_bytes, need);
  float* col = reinterpret_cast<float*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  const int K = C * kh * kw, P = g.Ho * g.Wo;
  const int Fg = F / num_group, Kg = K / num_group;
  hipStream_t st = (hipStream_t)stream;
  if (req_bias != SD_REQ_NULL) {
    SD_REQUIRE(d_bias, "d_bias is null");
    hipLaunchKernelGGL(dcn_bias_grad_kernel, dim3(F), dim3(256), 0, st, out_grad, d_bias, N, F, P,
                       req_bias == SD_REQ_ADD ? 1 : 0);
    SD_LAUNCH_CHECK();
  }
  // operand maxima for the scaled fp16 split: {max|W|, max|dY|, max|x| >= max|col|}
  unsigned* amax = dcn_amax_slots(col, (size_t)N * K * P);
  // fixed-point col2im: N * dgroup weight-sum bounds inside the slack behind the col matrix (what is left of
  // its 512 bytes depends on how far the caller's pointer was from a 256-byte boundary)
  unsigned* wsum = nullptr;
  const long slack_words = ((const char*)workspace + workspace_bytes - (const char*)amax) / 4;
  if (4 + (long)N * dgroup <= slack_words) wsum = amax + 4;   // N * dgroup words
  SD_HIP_CHECK(hipMemsetAsync(amax, 0, 16, st));
  launch_absmax(absmax_seg(weight, 1, F * Kg, F * Kg, 0, 1, amax),
                absmax_seg(out_grad, (long)N * F, P, P, 0, 1, amax + 1),
                absmax_seg(x, (long)N * C, H * W, H * W, 0, 1, amax + 2), st);
  if (req_x != SD_REQ_NULL || req_offset != SD_REQ_NULL) {
    // dcol[n][grp] (K/G x P) = W[grp]^T (K/G x F/G) . dY[n][grp] (F/G x P)
    for (int q = 0; q < num_group; ++q)
      if (int e = gemm_f32_impl(1, 0, Kg, P, Fg, weight + (long)q * Fg * Kg, Kg, 0, out_grad + (long)q * Fg * P, P,
                                (long)F * P, col + (long)q * Kg * P, P, (long)K * P, N, 0, amax, stream))
        return e;
    if (int e = sd_deform_col2im_coord(col, x, offset, d_offset, req_offset, N, C, H, W, kh, kw, pad,
                                       pad, stride, stride, dil, dil, dgroup, stream))
      return e;
    if (int e = col2im_impl(col, offset, d_x, req_x, N, C, H, W, kh, kw, pad, pad, stride, stride, dil, dil,
                            dgroup, stream, wsum))
      return e;
  }
  if (req_weight != SD_REQ_NULL) {
    SD_REQUIRE(d_weight, "d_weight is null");
    if (fwd_col) {
      col = const_cast<float*>(fwd_col);  // read only below
    } else if (int e = sd_deform_im2col(x, offset, col, N, C, H, W, kh, kw, pad, pad, stride, stride,
                                        dil, dil, dgroup, stream)) {
      return e;
    }
    if (req_weight == SD_REQ_WRITE)
      SD_HIP_CHECK(hipMemsetAsync(d_weight, 0, sizeof(float) * (size_t)F * Kg, st));
    // dW[grp] (F/G x K/G) += sum_n dY[n][grp] (F/G x P) . col[n][grp]^T (P x K/G): images in grid.z, atomic accumulate
    for (int q = 0; q < num_group; ++q)
      if (int e = gemm_f32_impl(0, 1, Fg, Kg, P, out_grad + (long)q * Fg * P, P, (long)F * P, col + (long)q * Kg * P,
                                P, (long)K * P, d_weight + (long)q * Fg * Kg, Kg, 0, N, 2, amax + 1, stream))
        return e;
  }
  return SD_OK;
}

extern "C" int sd_deform_conv_bwd(const float* out_grad, const float* x, const float* offset,
                                  const float* weight, float* d_x, float* d_offset,
                                  float* d_weight, int req_x, int req_offset, int req_weight,
                                  int N, int C, int H, int W, int F, int kh, int kw, int pad,
                                  int stride, int dil, int dgroup, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  return deform_conv_bwd_impl(out_grad, x, offset, weight, nullptr, d_x, d_offset, d_weight, nullptr, req_x,
                              req_offset, req_weight, SD_REQ_NULL, N, C, H, W, F, kh, kw, pad, stride, dil,
                              dgroup, 1, workspace, workspace_bytes, stream);
}

extern "C" const float* sd_deform_conv_col_of_workspace(const void* fwd_workspace) {
  return reinterpret_cast<const float*>(((uintptr_t)fwd_workspace + 255) & ~(uintptr_t)255);
}

extern "C" int sd_deform_conv_bwd_cached(const float* out_grad, const float* x, const float* offset,
                                         const float* weight, const float* fwd_col, float* d_x,
                                         float* d_offset, float* d_weight, int req_x,
                                         int req_offset, int req_weight, int N, int C, int H, int W,
                                         int F, int kh, int kw, int pad, int stride, int dil,
                                         int dgroup, void* workspace, size_t workspace_bytes,
                                         void* stream) {
  SD_REQUIRE(fwd_col, "fwd_col is null (use sd_deform_conv_bwd)");
  SD_REQUIRE((const void*)fwd_col != sd_deform_conv_col_of_workspace(workspace),
             "the backward's workspace must not be the forward's (dcol would overwrite col)");
  return deform_conv_bwd_impl(out_grad, x, offset, weight, fwd_col, d_x, d_offset, d_weight, nullptr, req_x,
                              req_offset, req_weight, SD_REQ_NULL, N, C, H, W, F, kh, kw, pad, stride, dil,
                              dgroup, 1, workspace, workspace_bytes, stream);
}

extern "C" int sd_deform_convolution_bwd(const float* out_grad, const float* x, const float* offset,
                                         const float* weight, const float* fwd_col, float* d_x, float* d_offset,
                                         float* d_weight, float* d_bias, int req_x, int req_offset, int req_weight,
                                         int req_bias, int N, int C, int H, int W, int F, int kh, int kw, int pad,
                                         int stride, int dil, int dgroup, int num_group, void* workspace,
                                         size_t workspace_bytes, void* stream) {
  SD_REQUIRE(!fwd_col || (const void*)fwd_col != sd_deform_conv_col_of_workspace(workspace),
             "the backward's workspace must not be the forward's (dcol would overwrite col)");
  return deform_conv_bwd_impl(out_grad, x, offset, weight, fwd_col, d_x, d_offset, d_weight, d_bias, req_x,
                              req_offset, req_weight, req_bias, N, C, H, W, F, kh, kw, pad, stride, dil, dgroup,
                              num_group, workspace, workspace_bytes, stream);
}

