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
