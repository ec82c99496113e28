// Shared host/device helpers for the simpledet_amd HIP library (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

// error codes returned through the C ABI (include/simpledet_ops.h)
#define SD_OK 0
#define SD_ERR_INVALID_ARG (-1)
#define SD_ERR_UNSUPPORTED (-2)
#define SD_ERR_HIP (-3)
#define SD_ERR_WORKSPACE (-4)

namespace sd {

// thread-local last-error message, read through sd_last_error()
char* err_buf();
int fail(int code, const char* fmt, ...);
// names of the kernels an entry point launches (sd_last_dispatch(): bench.py labels its roofline with them)
void note_dispatch(const char* fmt, ...);

#define SD_REQUIRE(cond, ...)                                    \
  do {                                                           \
    if (!(cond)) return ::sd::fail(SD_ERR_INVALID_ARG, __VA_ARGS__); \
  } while (0)

#define SD_HIP_CHECK(expr)                                                              \
  do {                                                                                  \
    hipError_t e_ = (expr);                                                             \
    if (e_ != hipSuccess)                                                               \
      return ::sd::fail(SD_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                            \
  } while (0)

#define SD_LAUNCH_CHECK() SD_HIP_CHECK(hipGetLastError())

// Kernel-VARIANT knobs (A/B of correct implementations from bench.py / the tests; defaults are the
// shipped path).  Lock-free reads.  Knobs that switch parts of a kernel OFF for profiling (results
// are wrong) exist only in the -DSD_PROFILING build (tools/libsimpledet_ops_hip_prof.so, `make prof`)
// and are compiled out of the product library.
int tuning(const char* key, int dflt);
#ifdef SD_PROFILING
#define SD_PROF_TUNING(key, dflt) ::sd::tuning(key, dflt)
#define SD_ABLATE(args, bits) ((args).ablate & (bits))
#else
#define SD_PROF_TUNING(key, dflt) (dflt)
#define SD_ABLATE(args, bits) 0  // the ablated branches are not in the product code at all
#endif

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// mshadow_op::maximum / minimum semantics (a > b ? a : b), kept explicit for NaN parity
__host__ __device__ static inline float fmaxr(float a, float b) { return a > b ? a : b; }
__host__ __device__ static inline float fminr(float a, float b) { return a < b ? a : b; }
__host__ __device__ static inline int imaxr(int a, int b) { return a > b ? a : b; }
__host__ __device__ static inline int iminr(int a, int b) { return a < b ? a : b; }

constexpr int kWave = 64;      // CDNA wavefront
constexpr int kNumXCD = 8;     // MI355X: 8 XCDs, block b is dispatched to XCD b % 8

// Compiler-level ordering of the LDS accesses of the lanes of ONE wave (the hardware already
// executes a wave's LDS instructions in order): without it a lane's loads may be hoisted above
// the stores other lanes issue in the same instruction stream.  Emits no instruction.
#if defined(__HIPCC__)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// Wave-wide reductions on the VALU only (DPP butterfly inside each row of 16 lanes, then the four
// row results through v_readlane): no LDS traffic, unlike __shfl_xor (ds_bpermute_b32).  The result
// is wave-uniform.  Every lane must be active.
#define SD_DPP_STEP(x, ctrl) __builtin_amdgcn_update_dpp((x), (x), (ctrl), 0xf, 0xf, false)
__device__ __forceinline__ float wave_max_f32(float v) {
  v = fmaxr(v, __int_as_float(SD_DPP_STEP(__float_as_int(v), 0xB1)));   // quad_perm [1,0,3,2]
  v = fmaxr(v, __int_as_float(SD_DPP_STEP(__float_as_int(v), 0x4E)));   // quad_perm [2,3,0,1]
  v = fmaxr(v, __int_as_float(SD_DPP_STEP(__float_as_int(v), 0x141)));  // row_half_mirror
  v = fmaxr(v, __int_as_float(SD_DPP_STEP(__float_as_int(v), 0x140)));  // row_mirror
  const int x = __float_as_int(v);
  const float r0 = __int_as_float(__builtin_amdgcn_readlane(x, 0));
  const float r1 = __int_as_float(__builtin_amdgcn_readlane(x, 16));
  const float r2 = __int_as_float(__builtin_amdgcn_readlane(x, 32));
  const float r3 = __int_as_float(__builtin_amdgcn_readlane(x, 48));
  return fmaxr(fmaxr(r0, r1), fmaxr(r2, r3));
}
__device__ __forceinline__ int wave_sum_i32(int v) {
  v += SD_DPP_STEP(v, 0xB1);
  v += SD_DPP_STEP(v, 0x4E);
  v += SD_DPP_STEP(v, 0x141);
  v += SD_DPP_STEP(v, 0x140);
  return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) +
         __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}

// max|.| of four gradients as the maximum of the BIT PATTERNS of |.| (non-negative floats order like unsigned
// integers; a NaN's pattern lies above inf's, so it survives every maximum -- `a > b ? a : b` on floats drops a NaN
// in its first operand, which let NaN gradients slip past the "non-finite -> float adds" test until round 6)
constexpr unsigned kFltMaxBits = 0x7f7fffffu;
__device__ __forceinline__ unsigned umaxr(unsigned a, unsigned b) { return a > b ? a : b; }
__device__ __forceinline__ unsigned absbits(float v) { return __float_as_uint(v) & 0x7fffffffu; }
__device__ __forceinline__ unsigned absbits4(const float4& v) {
  return umaxr(umaxr(absbits(v.x), absbits(v.y)), umaxr(absbits(v.z), absbits(v.w)));
}
__device__ __forceinline__ unsigned wave_max_u32(unsigned v) {
  v = umaxr(v, (unsigned)SD_DPP_STEP((int)v, 0xB1));
  v = umaxr(v, (unsigned)SD_DPP_STEP((int)v, 0x4E));
  v = umaxr(v, (unsigned)SD_DPP_STEP((int)v, 0x141));
  v = umaxr(v, (unsigned)SD_DPP_STEP((int)v, 0x140));
  const int x = (int)v;
  return umaxr(umaxr((unsigned)__builtin_amdgcn_readlane(x, 0), (unsigned)__builtin_amdgcn_readlane(x, 16)),
               umaxr((unsigned)__builtin_amdgcn_readlane(x, 32), (unsigned)__builtin_amdgcn_readlane(x, 48)));
}

// inclusive prefix sum over the 64 lanes on the VALU: row_shr 1 / 2 / 4 / 8 inside each row of 16 lanes (a lane
// whose source lies before its row keeps 0), then row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2
// and 3.  Every lane must be active.
__device__ __forceinline__ int wave_scan_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);
  return v;
}

// 32-bit fixed-point LDS sums have ONE unit per workgroup: 2^-30 .. 2^-29 of (max|g| x weight bound).  That is
// fine while the gradients of a workgroup span a few octaves; under heavy tails -- one element 10^4 x the
// typical one -- the typical elements would be rounded to a few units.  So every fixed-point backward looks at
// the dynamic range it actually streams and keeps the integer adds only while the unit is below ~2^-13 of the
// GEOMETRIC mean of its non-zero gradients (the mean itself is useless here: a lognormal's mean sits far out in
// its tail).  In exponents, all integer and therefore independent of any summation order:
//     E(max|g|) + ceil(log2(weight bound)) <= mean E(g) + kFxRangeBits - 1  (E = biased fp32 exponent field),
// the mean taken over a fixed sample of the stream (the first element of a 16-byte item; the band kernel looks at
// every fourth of its trips), zeros and denormals not counted.  Near-Gaussian gradients: E(max) - mean E ~ 3.5-4, so weight bounds up to ~2^11 pass;
// a loss scale cancels out.  A workgroup that fails takes the fp32 compare-and-swap adds -- the reference's own
// arithmetic (roi_align_v2.cu:67-83, upstream deformable_col2im: a float atomicAdd per tap).
constexpr int kFxRangeBits = 16;   // (the threshold below adds one bit for the optimistic maximum: 15 against the true one)
__device__ __forceinline__ int fp32_exponent_field(float v) { return (int)((__float_as_uint(v) >> 23) & 255u); }
__device__ __forceinline__ int ceil_log2_i32(int b) { return b <= 1 ? 0 : 32 - __builtin_clz((unsigned)(b - 1)); }
// The statistic travels as ONE integer per thread / wave / workgroup (one wave reduction, one LDS atomic, one LDS
// read behind the barrier that ends the scatter anyway -- every instruction in a short-lived workgroup's chain
// counts: two reductions and a 64-bit verdict cost the headline backward 2 us): per non-zero sample
//     ((E(g) - e_thr) << 12) + 1,    e_thr = E(optimistic max|g|) + 1 + ceil(log2(weight bound)) - kFxRangeBits,
// so the upper bits sum the (signed) margins and the low 12 bits count the samples (< 4096 per workgroup: the
// callers thin their sampling accordingly).  Fine <=> the margins sum to >= 0; when the true maximum turns out more
// than twice the optimistic one the margins are corrected by the difference of the exponents times the count.
__device__ __forceinline__ int fx_range_thr(float gmax_used, int bound) {
  return fp32_exponent_field(gmax_used) + 1 + ceil_log2_i32(bound) - kFxRangeBits;
}
__device__ __forceinline__ int fx_range_sample(float v, int e_thr) {
  const int ex = fp32_exponent_field(v);
  return ex ? ((ex - e_thr) << 12) + 1 : 0;
}
__device__ __forceinline__ bool fx_range_fine(int packed, float gmax_used, float gmax_true) {
  const int cnt = packed & 4095, margins = packed >> 12;
  const int extra = fp32_exponent_field(gmax_true) - (fp32_exponent_field(gmax_used) + 1);
  return margins - (extra > 0 ? extra * cnt : 0) >= 0;
}
// trips between samples (a power of two minus one) so that `per_trip` samples per trip over `trips` trips stay < 4096
__device__ __forceinline__ int fx_range_stride_mask(long per_trip, long trips, int at_least = 0) {
  int m = at_least;
  while (per_trip * ((trips + m) / (m + 1)) > 4000) m = 2 * m + 1;
  return m;
}

// fp32 add into an LDS word by compare-and-swap.  ds_add_f32 runs at ~0.33 lane-ops/clk/CU on
// gfx950 against ~2 for this loop and 4-9 for the integer LDS atomics (tools/lds_atomic_bench.hip,
// tools/lds_scatter_bench.hip), so every gradient plane kept in LDS is accumulated this way.
__device__ __forceinline__ void lds_add_cas(float* p, float v) {
  int* ip = reinterpret_cast<int*>(p);
  int old = *ip;
  while (true) {
    const int assumed = old;
    old = atomicCAS(ip, assumed, __float_as_int(__int_as_float(assumed) + v));
    if (old == assumed) break;
  }
}
#endif
constexpr int kNumCU = 256;

}  // namespace sd
