// _contrib_GenAnchor for gfx950: base anchors in double on the host (A <= 64 boxes), one 16-byte
// store per anchor on the device.
//   reference: operator_cxx/contrib/generate_anchor-inl.h:140-181 (base anchors, ratio-major),
//              generate_anchor.cu:61-81 (AnchorGridKernel: double add, cast to fp32 last),
//              symbol/builder.py:904-938 (numpy twin with the same grid layout)
// The op is a pure 16 B/anchor write (4.27 MB over all FPN levels): lanes own consecutive anchors
// so a wave stores 1 KiB contiguous; the reference's cudaMemcpy H2D of the base anchors and its
// default-stream launch are replaced by kernel arguments on the caller's stream.
#include "common.h"
#include "../../include/simpledet_ops.h"
#include <math.h>

namespace sd {

constexpr int kMaxBaseAnchors = 64;

struct AnchorArgs {
  double base[kMaxBaseAnchors * 4];
  float* out;
  int A, H, W, stride;
  long count;
};

__global__ __launch_bounds__(256) void anchor_grid_kernel(AnchorArgs a) {
  for (long index = (long)blockIdx.x * blockDim.x + threadIdx.x; index < a.count;
       index += (long)gridDim.x * blockDim.x) {
    const int an = (int)(index % a.A);
    const int w = (int)((index / a.A) % a.W);
    const int h = (int)(index / a.A / a.W);
    const double sx = (double)(w * a.stride), sy = (double)(h * a.stride);
    float4 v;
    v.x = (float)(a.base[an * 4 + 0] + sx);
    v.y = (float)(a.base[an * 4 + 1] + sy);
    v.z = (float)(a.base[an * 4 + 2] + sx);
    v.w = (float)(a.base[an * 4 + 3] + sy);
    reinterpret_cast<float4*>(a.out)[index] = v;
  }
}

// All pyramid levels in one launch (the reference runs one GenAnchor node per level,
// models/FPN/builder.py; five launches of a 2.7 us kernel are launch bound): level = blockIdx.y.
constexpr int kMaxLevelAnchors = 12;  // base anchors per level held in the kernel arguments
struct AnchorLevelsArgs {
  double base[SD_MAX_FPN_LEVELS][kMaxLevelAnchors * 4];
  float* out[SD_MAX_FPN_LEVELS];
  int H[SD_MAX_FPN_LEVELS], W[SD_MAX_FPN_LEVELS], stride[SD_MAX_FPN_LEVELS];
  int A, nlvl;
};

__global__ __launch_bounds__(256) void anchor_grid_levels_kernel(AnchorLevelsArgs a) {
  const int l = blockIdx.y;
  const long count = (long)a.H[l] * a.W[l] * a.A;
  for (long index = (long)blockIdx.x * blockDim.x + threadIdx.x; index < count;
       index += (long)gridDim.x * blockDim.x) {
    const int an = (int)(index % a.A);
    const int w = (int)((index / a.A) % a.W[l]);
    const int h = (int)(index / a.A / a.W[l]);
    const double sx = (double)(w * a.stride[l]), sy = (double)(h * a.stride[l]);
    float4 v;
    v.x = (float)(a.base[l][an * 4 + 0] + sx);
    v.y = (float)(a.base[l][an * 4 + 1] + sy);
    v.z = (float)(a.base[l][an * 4 + 2] + sx);
    v.w = (float)(a.base[l][an * 4 + 3] + sy);
    reinterpret_cast<float4*>(a.out[l])[index] = v;
  }
}

// generate_anchor-inl.h:140-181 with DType = double (generate_anchor.cu:122-133)
static void base_anchors(int feature_stride, const double* scales, int ns, const double* ratios,
                         int nr, double* base) {
  const double b0 = 0.0f, b1 = 0.0f;
  const double b2 = (double)((float)feature_stride - 1.0f);
  const double b3 = (double)((float)feature_stride - 1.0f);
  int n = 0;
  for (int j = 0; j < nr; ++j)
    for (int k = 0; k < ns; ++k) {
      const double scale = scales[k], ratio = ratios[j];
      const double w = b2 - b0 + 1.0f, h = b3 - b1 + 1.0f;
      const double x_ctr = b0 + 0.5 * (w - 1.0f), y_ctr = b1 + 0.5 * (h - 1.0f);
      const double size_ratios = (w * h) / ratio;
      const double new_w = rint(sqrt(size_ratios)) * scale;
      const double new_h = rint((new_w / scale * ratio)) * scale;
      base[n * 4 + 0] = x_ctr - 0.5f * (new_w - 1.0f);
      base[n * 4 + 1] = y_ctr - 0.5f * (new_h - 1.0f);
      base[n * 4 + 2] = x_ctr + 0.5f * (new_w - 1.0f);
      base[n * 4 + 3] = y_ctr + 0.5f * (new_h - 1.0f);
      ++n;
    }
}

}  // namespace sd

extern "C" int sd_gen_anchor(float* out, int H, int W, int feature_stride,
                             const double* scales_host, int n_scales, const double* ratios_host,
                             int n_ratios, void* stream) {
  using namespace sd;
  SD_REQUIRE(H >= 0 && W >= 0, "negative feature size");
  SD_REQUIRE(feature_stride > 0, "feature_stride must be positive");
  SD_REQUIRE(scales_host && ratios_host && n_scales > 0 && n_ratios > 0, "empty scales/ratios");
  SD_REQUIRE(n_scales * n_ratios <= kMaxBaseAnchors, "more than %d anchors per location",
             kMaxBaseAnchors);
  for (int i = 0; i < n_ratios; ++i) SD_REQUIRE(ratios_host[i] > 0, "ratio must be positive");
  AnchorArgs a;
  a.A = n_scales * n_ratios;
  base_anchors(feature_stride, scales_host, n_scales, ratios_host, n_ratios, a.base);
  a.out = out;
  a.H = H; a.W = W; a.stride = feature_stride;
  a.count = (long)H * W * a.A;
  if (a.count == 0) return SD_OK;
  SD_REQUIRE(out, "out is null");
  SD_REQUIRE(((uintptr_t)out & 15) == 0, "out must be 16-byte aligned");
  const int grid = (int)((a.count + 255) / 256 < kNumCU * 8 ? (a.count + 255) / 256 : kNumCU * 8);
  hipLaunchKernelGGL(anchor_grid_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  SD_LAUNCH_CHECK();
  return SD_OK;
}

extern "C" int sd_gen_anchor_levels(float* const* outs_host, const int* Hs_host, const int* Ws_host,
                                    const int* strides_host, int nlvl, const double* scales_host,
                                    int n_scales, const double* ratios_host, int n_ratios,
                                    void* stream) {
  using namespace sd;
  SD_REQUIRE(nlvl >= 0 && nlvl <= SD_MAX_FPN_LEVELS, "nlvl=%d outside [0,%d]", nlvl, SD_MAX_FPN_LEVELS);
  if (nlvl == 0) return SD_OK;
  SD_REQUIRE(outs_host && Hs_host && Ws_host && strides_host, "null level table");
  SD_REQUIRE(scales_host && ratios_host && n_scales > 0 && n_ratios > 0, "empty scales/ratios");
  const int A = n_scales * n_ratios;
  if (A > kMaxLevelAnchors) {  // many anchors per location: one launch per level
    for (int l = 0; l < nlvl; ++l)
      if (int e = sd_gen_anchor(outs_host[l], Hs_host[l], Ws_host[l], strides_host[l], scales_host,
                                n_scales, ratios_host, n_ratios, stream))
        return e;
    return SD_OK;
  }
  for (int i = 0; i < n_ratios; ++i) SD_REQUIRE(ratios_host[i] > 0, "ratio must be positive");
  AnchorLevelsArgs a{};
  a.A = A;
  a.nlvl = nlvl;
  long most = 0;
  for (int l = 0; l < nlvl; ++l) {
    SD_REQUIRE(Hs_host[l] >= 0 && Ws_host[l] >= 0, "negative feature size");
    SD_REQUIRE(strides_host[l] > 0, "feature_stride must be positive");
    const long count = (long)Hs_host[l] * Ws_host[l] * A;
    SD_REQUIRE(count == 0 || (outs_host[l] && ((uintptr_t)outs_host[l] & 15) == 0),
               "level %d: out must be a 16-byte aligned device pointer", l);
    base_anchors(strides_host[l], scales_host, n_scales, ratios_host, n_ratios, a.base[l]);
    a.out[l] = outs_host[l];
    a.H[l] = Hs_host[l]; a.W[l] = Ws_host[l]; a.stride[l] = strides_host[l];
    if (count > most) most = count;
  }
  if (most == 0) return SD_OK;
  const int gx = (int)((most + 255) / 256 < kNumCU * 8 ? (most + 255) / 256 : kNumCU * 8);
  hipLaunchKernelGGL(anchor_grid_levels_kernel, dim3(gx, nlvl), dim3(256), 0, (hipStream_t)stream, a);
  SD_LAUNCH_CHECK();
  return SD_OK;
}
