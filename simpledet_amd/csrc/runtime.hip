// Error reporting, ABI version and tuning knobs of libsimpledet_ops_hip.so.
#include "common.h"
#include <stdarg.h>
#include "../../include/simpledet_ops.h"
#include <atomic>
#include <string.h>
#include <hip/hip_fp16.h>

namespace sd {

char* err_buf() {
  static thread_local char buf[512] = {0};
  return buf;
}

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(err_buf(), 512, fmt, ap);
  va_end(ap);
  return code;
}

namespace {
struct Knob {
  const char* key;
  std::atomic<int> value;
  std::atomic<bool> set;
  Knob(const char* k, int v, bool s) : key(k), value(v), set(s) {}
};
// every knob a kernel launcher reads must be listed here (sd_set_tuning rejects unknown keys)
Knob g_knobs[] = {
    // ---- product: every key selects a kernel that some shape or entry point reaches without it (the
    // workspace-free entry points, C % 4 != 0, planes over the LDS budget, ...); tests force them here ----
    {"roi_align_fwd", 0, false},         // 1 (default): band-resident kernel / tiled kernels where it does not apply; 0: naive per-element kernel
    {"roi_align_fwd_band", 0, false},    // 1 (default) band-resident forward: planes streamed through LDS, no gathers; 0: the tiled fallback kernels
    {"roi_align_fwd_quad", 0, false},    // 1 (default): single-level float arg-max forward with four channel-last planes per workgroup when they fit (the C4 family); 0: the band kernel
    {"roi_align_bwd", 0, false},         // 2 (default) fused all-level kernel; 1 per-level LDS planes; 0 global atomics
    {"roi_align_bwd_flt4", 0, false},    // 1 (default): single-level float arg-max backward with four channel planes per workgroup when they fit
    {"roi_align_bwd_fx", 0, false},      // 1 (default): band sums in 32-bit fixed point (bit-reproducible); 0: fp32 compare-and-swap adds, hardware order
    {"roi_align_bwd_pixbound", 0, false}, // 1 (default): the list pre-pass bounds the weight per 4 x 4 pixel cell of a band (2-D difference array); 0: summed over the band's RoIs (what workspace-free calls get)
    {"roi_align_bwd_lists", 0, false},   // workspace pre-pass: 1 RoI lists + tap tables per band unit (default), 2 lists only, 0 none
    {"roi_pool_fwd", 0, false},          // 1 (default) four planes in LDS per workgroup, 2 one plane, 0 wave per (roi, channel)
    {"roi_pool_bwd", 0, false},          // 1 (default) LDS planes, four channels per workgroup, 2 one channel, 0 global atomics
    {"proposal_topk", 0, false},         // 0 by level size (default), 1 single workgroup, 2 multi-workgroup
    {"soft_nms_threads", 0, false},      // threads per problem: 64, 128 or 256 (default)
    {"deform_gemm_split", 0, false},     // 2 (default): scaled fp16 hi/lo split (needs operand maxima); 1: bf16 hi/lo split; 0: fp32 MFMA
    {"deform_gemm_vecstore", 0, false},  // 1 (default): whole-tile plain stores of the split GEMM leave as 512-byte tile rows through LDS; 0: element stores from the accumulator layout (what unaligned C gets)
    {"deform_gemm_nt", 0, false},        // 1 (default): those whole-tile stores are non-temporal; 0: plain
    {"deform_gemm_ksplit", 0, false},    // 1 (default): tiles of a mostly empty last round are cut into k slices (atomic adds)
    {"dcn_im2col", 0, false},            // 1 LDS-plane im2col (default), 0 per-lane global gathers
    {"dcn_window", 0, false},            // 1 stage only the touched range of each plane (default)
    {"dcn_im2col_pipe", 0, false},       // 1 (default): 3x3 im2col with the next window loads before this channel is sampled, col rows stored as 1 KB pieces behind them (s_waitcnt vmcnt(3)); 0: stores in place
    {"dcn_col2im", 0, false},            // 1 four channels per workgroup, shared sample geometry (default), 0 one channel
    {"dcn_col2im_fx", 0, false},         // 1 (default): the layer's backward sums dX in fixed point (integer LDS adds), 0 fp32 compare-and-swap
    {"dcn_coord", 0, false},             // 1 LDS-plane offset gradient (default), 0 per-lane gathers
    {"dcn_fused", 0, false},             // 1 (default): the col-free entry points sample inside the GEMM; 0: im2col + GEMM
    {"dcn_fused_tile", 0, false},        // pixels per tile of the fused kernels (1..96), 0 = balanced over the CUs (default): partial-tile tests
#ifdef SD_PROFILING
    // ---- profiling build only (tools/libsimpledet_ops_hip_prof.so): results are WRONG when != 0 ----
    {"roi_align_fwd_ablate", 0, false},
    {"roi_pool_fwd_ablate", 0, false},
    {"roi_align_bwd_ablate", 0, false},
    {"roi_align_dbg_lo", 0, false},      // device buffer for per-wave phase clocks
    {"roi_align_dbg_hi", 0, false},
    {"gemm_ablate", 0, false},           // skip the A (1) / B (2) prefetch of the split GEMM
    {"gemm_dbg_cap", 0, false},          // waves the buffer above has room for (split GEMM)
    {"dcn_im2col_nt", 0, false},         // bits 1-2: parts of the LDS im2col switched off
    {"dcn_fused_ablate", 0, false},      // parts of the fused forward switched off
#endif
};
}  // namespace

int tuning(const char* key, int dflt) {
  for (auto& k : g_knobs)
    if (!strcmp(k.key, key))
      return k.set.load(std::memory_order_acquire) ? k.value.load(std::memory_order_relaxed) : dflt;
  return dflt;
}

}  // namespace sd

namespace sd {
// HBM streaming copy: the measured-peak companion of the 8 TB/s spec figure and the known-byte
// calibration stream for rocprofv3's FETCH_SIZE / WRITE_SIZE (MI355X_MICROARCH.md, HBM).
template <typename T>
__global__ __launch_bounds__(256) void hbm_stream_copy(const T* __restrict__ src,
                                                       T* __restrict__ dst, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) dst[i] = src[i];
}
}  // namespace sd

extern "C" int sd_hbm_stream_copy(const void* src, void* dst, size_t bytes, int width_bytes,
                                  void* stream) {
  SD_REQUIRE(src && dst, "null buffer");
  SD_REQUIRE(width_bytes == 4 || width_bytes == 8 || width_bytes == 16, "width must be 4, 8, 16");
  SD_REQUIRE(bytes % width_bytes == 0, "bytes must be a multiple of the access width");
  hipStream_t st = (hipStream_t)stream;
  const size_t n = bytes / width_bytes;
  const int grid = sd::kNumCU * 8;
  if (width_bytes == 16)
    sd::hbm_stream_copy<float4><<<grid, 256, 0, st>>>((const float4*)src, (float4*)dst, n);
  else if (width_bytes == 8)
    sd::hbm_stream_copy<float2><<<grid, 256, 0, st>>>((const float2*)src, (float2*)dst, n);
  else
    sd::hbm_stream_copy<float><<<grid, 256, 0, st>>>((const float*)src, (float*)dst, n);
  SD_LAUNCH_CHECK();
  return SD_OK;
}

namespace sd {
// fp16 <-> fp32 streaming casts: the op-boundary casts of the fp16 RoIAlign backward (the backward
// kernel accumulates and writes fp32; its fp16 form wraps it in these two passes)
__global__ __launch_bounds__(256) void cast_h2f_kernel(const __half* __restrict__ src, float* __restrict__ dst, size_t n) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  const size_t stride = (size_t)gridDim.x * blockDim.x * 8;
  for (; i + 8 <= n; i += stride) {
    const uint4 v = *reinterpret_cast<const uint4*>(src + i);
    const __half2* h = reinterpret_cast<const __half2*>(&v);
    const float2 a = __half22float2(h[0]), b = __half22float2(h[1]), c = __half22float2(h[2]), d = __half22float2(h[3]);
    reinterpret_cast<float4*>(dst + i)[0] = make_float4(a.x, a.y, b.x, b.y);
    reinterpret_cast<float4*>(dst + i)[1] = make_float4(c.x, c.y, d.x, d.y);
  }
  if (i < n && i + 8 > n)
    for (size_t j = i; j < n; ++j) dst[j] = __half2float(src[j]);
}
__global__ __launch_bounds__(256) void cast_f2h_kernel(const float* __restrict__ src, __half* __restrict__ dst, size_t n, int add) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride)
    dst[i] = __float2half(add ? __half2float(dst[i]) + src[i] : src[i]);
}
}  // namespace sd

extern "C" int sd_cast_f16_to_f32(const void* src, float* dst, size_t n, void* stream) {
  if (n == 0) return SD_OK;
  SD_REQUIRE(src && dst, "null buffer");
  SD_REQUIRE(((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0, "buffers must be 16-byte aligned");
  sd::cast_h2f_kernel<<<sd::kNumCU * 8, 256, 0, (hipStream_t)stream>>>((const __half*)src, dst, n);
  SD_LAUNCH_CHECK();
  return SD_OK;
}

extern "C" int sd_cast_f32_to_f16(const float* src, void* dst, size_t n, int req, void* stream) {
  if (n == 0 || req == SD_REQ_NULL) return SD_OK;
  SD_REQUIRE(src && dst, "null buffer");
  SD_REQUIRE(req == SD_REQ_WRITE || req == SD_REQ_ADD, "req must be write or add");
  sd::cast_f2h_kernel<<<sd::kNumCU * 8, 256, 0, (hipStream_t)stream>>>(src, (__half*)dst, n, req == SD_REQ_ADD);
  SD_LAUNCH_CHECK();
  return SD_OK;
}

extern "C" int sd_stream_synchronize(void* stream) {
  SD_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream));
  return SD_OK;
}

namespace sd {
// the kernels the last RoIAlign entry point of this thread launched, for measurement tools
static thread_local char g_dispatch[256] = {0};
void note_dispatch(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_dispatch, sizeof(g_dispatch), fmt, ap);
  va_end(ap);
}
}  // namespace sd

extern "C" const char* sd_last_error(void) { return sd::err_buf(); }
extern "C" const char* sd_last_dispatch(void) { return sd::g_dispatch; }
extern "C" int sd_abi_version(void) { return SD_ABI_VERSION; }

extern "C" int sd_set_tuning(const char* key, int value) {
  if (!key) return sd::fail(SD_ERR_INVALID_ARG, "null tuning key");
  for (auto& k : sd::g_knobs)
    if (!strcmp(k.key, key)) {
      k.value.store(value, std::memory_order_relaxed);
      k.set.store(true, std::memory_order_release);
      return SD_OK;
    }
  return sd::fail(SD_ERR_INVALID_ARG, "unknown tuning key '%s'", key);
}

extern "C" int sd_get_tuning(const char* key, int* value) {
  if (!key || !value) return sd::fail(SD_ERR_INVALID_ARG, "null argument");
  for (auto& k : sd::g_knobs)
    if (!strcmp(k.key, key)) {
      *value = k.set.load() ? k.value.load() : -1;
      return SD_OK;
    }
  return sd::fail(SD_ERR_INVALID_ARG, "unknown tuning key '%s'", key);
}
