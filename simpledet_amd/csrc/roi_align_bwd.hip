// ROIAlign_v2 backward for gfx950 (MI355X): the fused all-level kernel (roi_align_bwd_packed4: packed or
// float arg-max, fp32 or fp16 I/O), the four-plane kernel of the C4 family (roi_align_bwd_flt4_kernel), and
// the per-level fallbacks (LDS planes, global atomics).  reference: operator_cxx/contrib/roi_align_v2.cu:35-84.
#include "roi_align_lists.h"

namespace sd {

// reference structure: zero-fill (by the caller) + 4 global atomics per output element
__global__ __launch_bounds__(256) void roi_align_bwd_atomic(BwdArgs a) {
  const long count = (long)a.B * a.R * a.C * a.PP;
  for (long index = (long)blockIdx.x * blockDim.x + threadIdx.x; index < count;
       index += (long)gridDim.x * blockDim.x) {
    const int c = (int)((index / a.PP) % a.C);
    const int n = (int)(index / a.PP / a.C);
    if (a.filter_lvl >= 0) {
      const float* r = a.rois + (long)n * 4;
      if (fpn_level(r[0], r[1], r[2], r[3], a.L) != a.filter_lvl) continue;
    }
    const float a_x = a.ax[index], a_y = a.ay[index];
    if (a_x != -1.f && a_y != -1.f) {
      const int H = a.H, W = a.W;
      float* d = a.dx + ((long)(n / a.R) * a.C + c) * H * W;
      int hlow = iminr(imaxr((int)floorf(a_y), 0), H - 1);
      int hhigh = iminr(imaxr((int)ceilf(a_y), 0), H - 1);
      int wleft = iminr(imaxr((int)floorf(a_x), 0), W - 1);
      int wright = iminr(imaxr((int)ceilf(a_x), 0), W - 1);
      float alpha = (hlow == hhigh) ? 0.5f : (a_y - (float)hlow) / (float)(hhigh - hlow);
      float beta = (wleft == wright) ? 0.5f : (a_x - (float)wleft) / (float)(wright - wleft);
      const float g = a.dy[index];
      atomicAdd(d + hlow * W + wleft, g * (1 - alpha) * (1 - beta));
      atomicAdd(d + hlow * W + wright, g * (1 - alpha) * beta);
      atomicAdd(d + hhigh * W + wleft, g * alpha * (1 - beta));
      atomicAdd(d + hhigh * W + wright, g * alpha * beta);
    }
  }
}

// Per-level plane kernel (knob roi_align_bwd = 1; also the fallback when the fused kernel does not
// apply).  64-bit fixed-point planes: a per-workgroup power-of-two scale chosen
// from max|dY| of the workgroup's own items (no overflow by construction), every tap value still
// computed in float exactly as the reference does, only the summation exact instead of
// float-in-arbitrary-order -- a bit-reproducible backward.  It is not faster (ds_add_u64 sustains
// no more adds than the CAS loop in this access pattern, tools/lds_scatter_bench.hip, and the
// planes take twice the LDS).  Non-finite dY (inf/nan must propagate) falls back to the CAS loop.
__device__ __forceinline__ void lds_add_fx(long long* p, float v, double scale) {
  const long long q = __double2ll_rn((double)v * scale);
  __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(p), (unsigned long long)q,
                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// fire-and-forget 32-bit fixed-point add (no return value, no dependent LDS round trip)
__device__ __forceinline__ void lds_add_i32(int* p, float v, float scale) {
  __hip_atomic_fetch_add(p, __float2int_rn(v * scale), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// The tap-table form of the headline backward (round 6: its workgroups are VALU bound, 69 % of the SIMD cycles,
// ~50 vector instructions per bin): the value is already scaled, the address is a byte offset into the band (the
// pre-pass stores its offsets times four), and the conversion is ONE instruction -- v_cvt_rpi_i32_f32,
// floor(x + 0.5), where __float2int_rn is v_rndne_f32 + v_cvt_i32_f32.  (Round half up instead of half to even: a
// tie is as far from either neighbour; the sums stay order independent.)
typedef __attribute__((address_space(3))) int lds_i32;
__device__ __forceinline__ unsigned lds_offset_of(const void* p) {   // LDS byte address of a __shared__ object
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
__device__ __forceinline__ void lds_add_i32_at(unsigned lds_byte_addr, float v) {
  int q;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(q) : "v"(v));
  __hip_atomic_fetch_add((lds_i32*)(uintptr_t)lds_byte_addr, q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// One workgroup owns CPB channel planes (rows [row0,row1) of them) of one image in LDS.
//   FX = true : int64 fixed-point planes (8 B per pixel), float-CAS fallback on non-finite dY (the only
//   instantiation; FX = false -- float planes with a CAS loop -- was the round-1 A/B twin)
template <int PP, int CPB, int THREADS, bool FX>
__global__ __launch_bounds__(THREADS) void roi_align_bwd_plane(BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int U = 4;  // items per lane per trip: 3*U independent global loads in flight
  constexpr int ESZ = FX ? 2 : 1;  // plane element size in floats
  const int tid = threadIdx.x;
  const int H = a.H, W = a.W;
  const int ncb = a.C / CPB;
  // block -> (unit = image x band, channel block); channel blocks of one unit are contiguous per
  // XCD so the dY/argmax lines two neighbouring channels share are fetched into one L2 only
  int u, cb;
  if (ncb % kNumXCD == 0) {
    const int xcd = blockIdx.x % kNumXCD, j = blockIdx.x / kNumXCD, per = ncb / kNumXCD;
    cb = xcd * per + (j % per);
    u = j / per;
  } else {
    cb = blockIdx.x % ncb;
    u = blockIdx.x / ncb;
  }
  const int img = u / a.nbands, band = u % a.nbands;
  const int row0 = band * a.band_rows;
  const int row1 = iminr(row0 + a.band_rows, H);
  const int band_elems = (row1 - row0) * W;  // per channel
  const int c0 = cb * CPB;

  // LDS: [plane: CPB*band_elems elements, padded to 4][RoI list: R ints][counter][gmax bits][flag]
  const int plane_total = CPB * band_elems;
  const int plane_pad = (plane_total + 3) & ~3;
  float* planef = smem;
  long long* planeq = reinterpret_cast<long long*>(smem);
  int* list = reinterpret_cast<int*>(smem + (size_t)plane_pad * ESZ);
  int* nlist = list + a.R;
  unsigned* gmax_bits = reinterpret_cast<unsigned*>(nlist + 1);
  int* nonfinite = nlist + 2;

  {
    float4* p4 = reinterpret_cast<float4*>(smem);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < plane_pad * ESZ / 4; i += THREADS) p4[i] = z;
  }
  if (tid == 0) {
    *nlist = 0;
    *gmax_bits = 0u;
    *nonfinite = 0;
  }
  __syncthreads();

  // ---- RoIs of this image that can touch this band (and belong to this level) ----
  for (int r = tid; r < (SD_ABLATE(a, 4) ? 0 : a.R); r += THREADS) {
    const float4 rb = *reinterpret_cast<const float4*>(a.rois + ((long)img * a.R + r) * 4);
    bool take = true;
    if (a.filter_lvl >= 0) take = fpn_level(rb.x, rb.y, rb.z, rb.w, a.L) == a.filter_lvl;
    if (take && a.nbands > 1) {
      // conservative row range of every tap of this RoI (taps lie within the clipped bins +-1)
      float s = fminr(fmaxr(rb.y * a.scale, 0.f), (float)(H - 1));
      float e = fminr(fmaxr(rb.w * a.scale, 0.f), (float)(H - 1));
      float lo = fminr(s, e) - 2.f, hi = fmaxr(s, e) + 2.f;
      if (hi < (float)row0 || lo > (float)(row1 - 1)) take = false;
    }
    if (take) list[atomicAdd(nlist, 1)] = r;
  }
  __syncthreads();
  int nitems = *nlist * (CPB * PP);
  if (SD_ABLATE(a, 1)) nitems = 0;

  const long roi_stride = (long)a.C * PP;
  const long img_base = (long)img * a.R * roi_stride + (long)c0 * PP;

  // ---- fixed-point scale: 2^S * (sum of |taps| on any pixel) < 2^62 ----
  double fx_scale = 0.0, fx_inv = 0.0;
  bool use_fx = FX;
  if (FX) {
    float m = 0.f;
    int bad = 0;
    for (int it = tid; it < nitems; it += THREADS) {
      const int li = it / (CPB * PP);
      const float g = a.dy[img_base + (long)list[li] * roi_stride + it % (CPB * PP)];
      const float ag = fabsf(g);
      bad |= !(ag <= FLT_MAX);
      m = fmaxr(m, ag);
    }
    if (bad) atomicOr(nonfinite, 1);
    atomicMax(gmax_bits, __float_as_uint(m));  // non-negative floats order like their bit patterns
    __syncthreads();
    const float gmax = __uint_as_float(*gmax_bits);
    if (*nonfinite) {
      use_fx = false;  // planes are zero in both representations
    } else if (gmax == 0.f) {
      nitems = 0;
    } else {
      int e;
      frexp((double)gmax * (double)(nitems / CPB), &e);  // bound < 2^e
      fx_scale = ldexp(1.0, 61 - e);
      fx_inv = ldexp(1.0, e - 61);
    }
  }

  // ---- scatter bins into the LDS planes ----
  for (int it0 = tid; it0 < nitems; it0 += U * THREADS) {
    float vx[U], vy[U], vg[U];
    int rem[U];
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const int it = it0 + k * THREADS;
      vx[k] = -1.f;
      if (it < nitems) {
        const int li = it / (CPB * PP);
        rem[k] = it % (CPB * PP);
        const long idx = img_base + (long)list[li] * roi_stride + rem[k];
        vx[k] = a.ax[idx];
        vy[k] = a.ay[idx];
        vg[k] = a.dy[idx];
      }
    }
#pragma unroll
    for (int k = 0; k < U; ++k) {
      const float a_x = vx[k], a_y = vy[k];
      if (a_x != -1.f && a_y != -1.f) {
        const float g = vg[k];
        int hlow = iminr(imaxr((int)floorf(a_y), 0), H - 1);
        int hhigh = iminr(imaxr((int)ceilf(a_y), 0), H - 1);
        int wleft = iminr(imaxr((int)floorf(a_x), 0), W - 1);
        int wright = iminr(imaxr((int)ceilf(a_x), 0), W - 1);
        float alpha = (hlow == hhigh) ? 0.5f : (a_y - (float)hlow) / (float)(hhigh - hlow);
        float beta = (wleft == wright) ? 0.5f : (a_x - (float)wleft) / (float)(wright - wleft);
        const int pb = (rem[k] / PP) * band_elems;
        const float w00 = g * (1 - alpha) * (1 - beta), w01 = g * (1 - alpha) * beta;
        const float w10 = g * alpha * (1 - beta), w11 = g * alpha * beta;
        const bool top = hlow >= row0 && hlow < row1, bot = hhigh >= row0 && hhigh < row1;
        const int o0 = pb + (hlow - row0) * W, o1 = pb + (hhigh - row0) * W;
        if (use_fx) {
          if (top) {
            lds_add_fx(planeq + o0 + wleft, w00, fx_scale);
            lds_add_fx(planeq + o0 + wright, w01, fx_scale);
          }
          if (bot) {
            lds_add_fx(planeq + o1 + wleft, w10, fx_scale);
            lds_add_fx(planeq + o1 + wright, w11, fx_scale);
          }
        } else {
          if (top) {
            lds_add_cas(planef + o0 + wleft, w00);
            lds_add_cas(planef + o0 + wright, w01);
          }
          if (bot) {
            lds_add_cas(planef + o1 + wleft, w10);
            lds_add_cas(planef + o1 + wright, w11);
          }
        }
      }
    }
  }
  __syncthreads();

  // ---- write the band out once.  With one band the CPB planes are contiguous in HBM; with
  // several bands CPB == 1 and rows [row0,row1) of one plane are contiguous ----
  if (SD_ABLATE(a, 2)) return;
  const long off = (((long)img * a.C + c0) * H + row0) * W;
  float* dst = a.dx + off;
  auto get = [&](int i) -> float {
    return use_fx ? (float)((double)planeq[i] * fx_inv) : planef[i];
  };
  if (((off | plane_total) & 3) == 0) {
    float4* d4 = reinterpret_cast<float4*>(dst);
    for (int i = tid; i < plane_total / 4; i += THREADS) {
      float4 v;
      if (use_fx) {
        const longlong2 q0 = reinterpret_cast<const longlong2*>(planeq)[2 * i];
        const longlong2 q1 = reinterpret_cast<const longlong2*>(planeq)[2 * i + 1];
        v = make_float4((float)((double)q0.x * fx_inv), (float)((double)q0.y * fx_inv),
                        (float)((double)q1.x * fx_inv), (float)((double)q1.y * fx_inv));
      } else {
        v = reinterpret_cast<const float4*>(planef)[i];
      }
      if (a.req == SD_REQ_ADD) {
        const float4 o = d4[i];
        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
      }
      d4[i] = v;
    }
  } else {
    for (int i = tid; i < plane_total; i += THREADS)
      dst[i] = (a.req == SD_REQ_ADD) ? dst[i] + get(i) : get(i);
  }
}

// ------------------------------------------------------------------------------------------------
// fused backward, wide-load kernel
// ------------------------------------------------------------------------------------------------
// Workgroup = (level, image, row band, channel), band of the gradient plane in LDS, written to HBM
// once.  The item loop is built around the NUMBER of vector-memory instructions (a wave64 load
// occupies the address unit ~16 clocks whatever its width; the round-1 kernel, removed in round 5,
// issued four loads per (RoI, bin) item -- arg-max byte, two 8-byte table entries, gradient -- 2.0 M
// wave instructions per launch).
// Here
//   * a lane owns FOUR consecutive bins of one RoI: one aligned 4-byte load brings their four
//     arg-max codes (rows are padded to whole dwords, amax_stride) and one 16-byte load the four
//     gradients: 0.5 load per item instead of 2;
//   * the per-RoI sample-coordinate tables (3*(PH+PW) floats each, written by the forward) of the
//     RoIs on the band are staged ONCE per workgroup into LDS with 8-byte loads; an item then picks
//     its row / column coordinate with two ds_read_b32 and derives the neighbours and the
//     interpolation fraction with the backward's own expressions (floor / ceil / clamp, v - low);
//   * the band is accumulated in 32-bit FIXED POINT with plain integer LDS atomics (ds_add_u32,
//     fire and forget: 3.5 adds/clk/CU against 1.9 for the float compare-and-swap loop whose two
//     dependent LDS round trips per add were the longest chain of the workgroup).  Every tap value
//     is still computed in fp32 exactly as the reference does; only the SUM is exact integer
//     arithmetic on values rounded to 2^-S, so the result does not depend on the order of the
//     adds: the backward is bit-reproducible from run to run.  S is chosen per workgroup from
//     max|dY| of its own items and a rigorous bound on how much one pixel can receive, so the sum
//     cannot overflow and one add is off by at most 2^-(S+1):
//         pixel sum <= max|dY| * sum over the band's RoIs of nx*ny,
//         nx = min(PW, floor(2 / bin width) + 2) = bins of the RoI whose sample can lie within one
//         pixel of a given column (a bin adds total weight <= 1), ny likewise;
//     at the baseline that is ~2e-6 * max|dY| per unit.  Non-finite dY (inf / nan must propagate)
//     switches the workgroup to the float compare-and-swap adds.
// HALF: the gradient comes in and the feature gradients go out as fp16 (the sums are formed exactly as
// in the fp32 kernel: fp32 tap values, fixed-point or fp32 accumulation in LDS; only the two I/O
// conversions move into the kernel) -- what an fp16 graph computes with the reference's casts around
// the op (models/FPN/builder.py:581-586, 607-608), without the two cast passes over 26 + 182 MB.
struct __attribute__((packed, aligned(2))) H4u {
  __half x, y, z, w;
};
template <int PH, int PW, int THREADS, int TCH, int MODE, bool HALF = false>
__global__ __launch_bounds__(THREADS) __attribute__((amdgpu_waves_per_eu(8, 8)))  // <= 64 VGPRs: four 512-thread workgroups per CU
void roi_align_bwd_packed4(BwdFusedArgs a) {
  using TIO = typename std::conditional<HALF, __half, float>::type;
  static_assert(!HALF || MODE != 2, "fp16 I/O goes with the packed arg-max");
  // MODE 0: one-byte arg-max codes + the forward's coordinate table (4 bytes per sample coordinate);
  // MODE 1 (TAPS): the workspace pre-pass has left band-relative tap entries (8 bytes each) for the
  // listed RoIs; MODE 2 (FLT): the reference's float arg-max planes (the drop-in ROIAlign_v2 op and
  // the three-output fused op): an item carries its four (x, y) coordinates, no table
  constexpr bool TAPS = MODE == 1, FLT = MODE == 2;
  constexpr int PP = PH * PW, PPS = amax_stride(PP), GP = (PP + 3) / 4, NE = 3 * (PH + PW);
  constexpr int TS = TAPS ? 2 * NE : NE;
  constexpr int CW = kCoordWords * (PH + PW);  // words per RoI in the forward's table
  constexpr bool TAIL = (PP % 4) != 0;         // last lane of a RoI owns fewer than four bins
  static_assert(TS % 4 == 0 || !TAPS, "tap tables are copied as float4");
  static_assert(TS % 2 == 0, "table rows are copied as float2");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  // ---- block -> (level, image, band, channel) ----
  int li = 0;
  while (li + 1 < a.nlaunch && (int)blockIdx.x >= a.block_end[li]) ++li;
  const int lvl = a.order[li];
  const int b0 = (int)blockIdx.x - (li ? a.block_end[li - 1] : 0);
  const int H = a.L.H[lvl], W = a.L.W[lvl];
  const int nbands = a.nbands[lvl];
  int u, c;
  if (a.C % kNumXCD == 0) {  // an XCD keeps a contiguous channel range: dY/argmax lines stay in one L2
    const int xcd = b0 % kNumXCD, j = b0 / kNumXCD, per = a.C / kNumXCD;
    c = xcd * per + (j % per);
    u = j / per;
  } else {
    c = b0 % a.C;
    u = b0 / a.C;
  }
  const int img = u / nbands, band = u % nbands;
  const int row0 = band * a.band_rows[lvl];
  const int row1 = iminr(row0 + a.band_rows[lvl], H);
  const int band_elems = (row1 - row0) * W;
  const int plane_pad = (band_elems + 3) & ~3;
  float* plane = smem;
  float* tab = smem + plane_pad;                      // [TCH][TS] sample coordinates
  int* list = reinterpret_cast<int*>(tab + (FLT ? 0 : TCH * TS));  // RoIs of this image on this band
  int* nlist = list + a.R;  // [0] count [1] bound [2] max|dY| bits, first chunk [3] non-finite [4] max|dY| bits, all
  int* plane_i = reinterpret_cast<int*>(smem);
  const unsigned plane_lds = lds_offset_of(smem);

  // the RoI boxes (or the unit's list) are in flight while the band is zeroed
  int* wcnt = nlist + 8;  // [THREADS / 64] scratch of the list builder
  const int* wl = nullptr;
  int wl_n = 0, wl_bound = 0, wl_first = 0;
  float4 rb0 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (a.ws_list) {
    wl = a.ws_list + (long)(a.unit_base[li] + u) * (a.R + 2);
    wl_n = wl[0];
    wl_bound = wl[1];
    if (tid < wl_n) wl_first = wl[2 + tid];
  } else if (tid < a.R) {
    rb0 = *reinterpret_cast<const float4*>(a.rois + ((long)img * a.R + tid) * 4);
  }
  {
    float4* p4 = reinterpret_cast<float4*>(smem);
    const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int i = tid; i < plane_pad / 4; i += THREADS) p4[i] = z;
  }
  if (tid < 8) nlist[tid] = 0;
  if (wl) {
    if (tid < wl_n) list[tid] = wl_first;
    for (int i = tid + THREADS; i < wl_n; i += THREADS) list[i] = wl[2 + i];
    __syncthreads();  // (nlist cleared)
    if (tid == 0) {
      nlist[0] = wl_n;
      nlist[1] = wl_bound;
    }
    __syncthreads();
  } else {
    __syncthreads();
    bwd_band_list<PH, PW, THREADS>(a, lvl, img, nbands, row0, row1, rb0, list, nlist, wcnt);
  }
  const int nl = *nlist;

  // wave-uniform bases + 32-bit lane offsets (the launcher checks R*C*PP < 2^31)
  const int roi_stride = a.C * PP;
  const TIO* dyb = reinterpret_cast<const TIO*>(a.dy) + (long)img * a.R * roi_stride + (long)c * PP;
  const float* axb = FLT ? a.ax + (long)img * a.R * roi_stride + (long)c * PP : nullptr;
  const float* ayb = FLT ? a.ay + (long)img * a.R * roi_stride + (long)c * PP : nullptr;
  const unsigned char* amb = FLT ? nullptr : a.amax8 + ((long)img * a.R * a.C + c) * PPS;
  const int am_stride = a.C * PPS;
  const float* cob = a.coords + (long)img * a.R * CW;

  bool use_fx = !a.float_adds;  // (`roi_align_bwd_fx` = 0: fp32 compare-and-swap adds in every workgroup)
  float fx_scale = 1.f, fx_inv = 1.f;
  struct Item {
    float4 g;       // gradients of bins b0 .. b0+3
    float4 x, y;    // FLT: their arg-max coordinates (-1: nothing pooled)
    unsigned code;  // their four arg-max codes, one per byte (255: nothing pooled); FLT: 0 = item present
    int j, b0;      // RoI slot (in the list / in the streamed chunk), first bin
  };
  auto load_item = [&](int t, int cb, int nli, Item& it) {
    it.code = 0xffffffffu;
    it.j = 0;
    it.b0 = 0;
    it.g = make_float4(0.f, 0.f, 0.f, 0.f);
    it.x = it.y = make_float4(-1.f, -1.f, -1.f, -1.f);
    if (t < nli) {
      const int j = t / GP, g = t - j * GP;
      const int r = list[cb + j];
      it.j = j;
      it.b0 = 4 * g;
      unsigned code = FLT ? 0u : *reinterpret_cast<const unsigned*>(amb + r * am_stride + 4 * g);
      // bins PP-4 .. PP-1 are fetched by the last lane of a RoI, only the last PP % 4 belong to it
      constexpr int KEEP = TAIL ? PP % 4 : 1;
      // four gradients at element offset o (fp16: 8 bytes, converted)
      auto load_g = [&](int o) {
        if constexpr (HALF) {
          const H4u h = *reinterpret_cast<const H4u*>(dyb + o);
          F4u v;
          v.x = __half2float(h.x); v.y = __half2float(h.y); v.z = __half2float(h.z); v.w = __half2float(h.w);
          return v;
        } else {
          return *reinterpret_cast<const F4u*>(dyb + o);
        }
      };
      auto tail4 = [](const F4u& v, float fill) {
        const float gg[4] = {v.x, v.y, v.z, v.w};
        return make_float4(gg[4 - KEEP], KEEP > 1 ? gg[KEEP > 1 ? 5 - KEEP : 0] : fill,
                           KEEP > 2 ? gg[KEEP > 2 ? 6 - KEEP : 0] : fill, fill);
      };
      if (TAIL && g == GP - 1) {
        const int o = r * roi_stride + (PP - 4);
        it.g = tail4(load_g(o), 0.f);
        if (FLT) {
          it.x = tail4(*reinterpret_cast<const F4u*>(axb + o), -1.f);
          it.y = tail4(*reinterpret_cast<const F4u*>(ayb + o), -1.f);
        }
        code |= FLT ? 0u : 0xffffffffu << (8 * KEEP);  // the padding bytes of the row are not codes
      } else {
        const int o = r * roi_stride + 4 * g;
        const F4u v = load_g(o);
        it.g = make_float4(v.x, v.y, v.z, v.w);
        if (FLT) {
          const F4u vx = *reinterpret_cast<const F4u*>(axb + o), vy = *reinterpret_cast<const F4u*>(ayb + o);
          it.x = make_float4(vx.x, vx.y, vx.z, vx.w);
          it.y = make_float4(vy.x, vy.y, vy.z, vy.w);
        }
      }
      it.code = code;
    }
  };
  auto scatter_item = [&](const Item& it, int slot) {
    if (it.code == 0xffffffffu || (SD_ABLATE(a, 1))) return;  // (profiling build, 1: no scatter)
    const float* tj = tab + slot * TS;
    const float gg[4] = {it.g.x, it.g.y, it.g.z, it.g.w};
    const float xx[4] = {it.x.x, it.x.y, it.x.z, it.x.w}, yy[4] = {it.y.x, it.y.y, it.y.z, it.y.w};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float a_x, a_y;
      if (FLT) {
        a_x = xx[s];
        a_y = yy[s];
        if (a_x == -1.f || a_y == -1.f) continue;  // roi_align_v2.cu:53: nothing was pooled
      } else {
        const int code = (it.code >> (8 * s)) & 0xff;
        if (code == 255) continue;
        const int bin = it.b0 + s;
        const int p = bin / PW, q = bin - p * PW;
        const int k = (code * 11) >> 5, l = code - 3 * k;  // code = 3k + l, k,l in 0..2
        a_y = tj[p * 3 + k];
        a_x = tj[3 * PH + q * 3 + l];
      }
      const int hlow = iminr(imaxr((int)floorf(a_y), 0), H - 1);
      const int hhigh = iminr(imaxr((int)ceilf(a_y), 0), H - 1);
      const int wleft = iminr(imaxr((int)floorf(a_x), 0), W - 1);
      const int wright = iminr(imaxr((int)ceilf(a_x), 0), W - 1);
      // (v - low) / (high - low) with high - low == 1
      const float alpha = (hlow == hhigh) ? 0.5f : (a_y - (float)hlow);
      const float beta = (wleft == wright) ? 0.5f : (a_x - (float)wleft);
      const float g = gg[s];
      const float w00 = g * (1 - alpha) * (1 - beta), w01 = g * (1 - alpha) * beta;
      const float w10 = g * alpha * (1 - beta), w11 = g * alpha * beta;
      const int o0 = (hlow - row0) * W, o1 = (hhigh - row0) * W;
      if (use_fx) {
        if (hlow >= row0 && hlow < row1) {
          lds_add_i32(plane_i + o0 + wleft, w00, fx_scale);
          lds_add_i32(plane_i + o0 + wright, w01, fx_scale);
        }
        if (hhigh >= row0 && hhigh < row1) {
          lds_add_i32(plane_i + o1 + wleft, w10, fx_scale);
          lds_add_i32(plane_i + o1 + wright, w11, fx_scale);
        }
        continue;
      }
      if (hlow >= row0 && hlow < row1) {
        lds_add_cas(plane + o0 + wleft, w00);
        lds_add_cas(plane + o0 + wright, w01);
      }
      if (hhigh >= row0 && hhigh < row1) {
        lds_add_cas(plane + o1 + wleft, w10);
        lds_add_cas(plane + o1 + wright, w11);
      }
    }
  };

  // TAPS: a bin reads the two 8-byte tap entries of its winning sample (row entry: band offsets of
  // the two neighbour rows, 0xffff = outside the band, + fraction; column entry: the two columns +
  // fraction) and needs no floor / ceil / clamp / row multiply of its own; with fixed point the
  // gradient is scaled first (a power of two: the products are the same floats times 2^S).
  auto scatter_taps = [&](const Item& it, int slot) {
    if (it.code == 0xffffffffu || (SD_ABLATE(a, 1))) return;  // (profiling build, 1: no scatter)
    const float* tj = tab + slot * TS;
    const float gg[4] = {it.g.x, it.g.y, it.g.z, it.g.w};
    int p = it.b0 / PW, q = it.b0 - p * PW;
#pragma unroll
    for (int s = 0; s < 4; ++s, ++q) {
      if (q == PW) { q = 0; ++p; }
      const int code = (it.code >> (8 * s)) & 0xff;
      if (code == 255) continue;
      const int k = __mul24(code, 11) >> 5, l = code - __mul24(k, 3);  // code = 3k + l, k,l in 0..2
      const float2 re = *reinterpret_cast<const float2*>(tj + 2 * (__mul24(p, 3) + k));
      const float2 ce = *reinterpret_cast<const float2*>(tj + 2 * (3 * PH + __mul24(q, 3) + l));
      // (the entries hold BYTE offsets into the band: row offsets 4 * (row - row0) * W, 0xffff = outside; columns 4 * col)
      const unsigned rp = __float_as_uint(re.x), cp = __float_as_uint(ce.x);
      const unsigned o0 = rp & 0xffffu, o1 = rp >> 16, wleft = cp & 0xffffu, wright = cp >> 16;
      const float alpha = re.y, beta = ce.y;
      const float g = use_fx ? gg[s] * fx_scale : gg[s];
      const float w00 = g * (1 - alpha) * (1 - beta), w01 = g * (1 - alpha) * beta;
      const float w10 = g * alpha * (1 - beta), w11 = g * alpha * beta;
      if (use_fx) {
        if (o0 != 0xffff) {
          lds_add_i32_at(plane_lds + o0 + wleft, w00);
          lds_add_i32_at(plane_lds + o0 + wright, w01);
        }
        if (o1 != 0xffff) {
          lds_add_i32_at(plane_lds + o1 + wleft, w10);
          lds_add_i32_at(plane_lds + o1 + wright, w11);
        }
        continue;
      }
      if (o0 != 0xffff) {
        lds_add_cas(plane + ((o0 + wleft) >> 2), w00);
        lds_add_cas(plane + ((o0 + wright) >> 2), w01);
      }
      if (o1 != 0xffff) {
        lds_add_cas(plane + ((o1 + wleft) >> 2), w10);
        lds_add_cas(plane + ((o1 + wright) >> 2), w11);
      }
    }
  };

  auto stage_tables = [&](int cb, int ncur) {
    if (FLT) return;
    if (TAPS) {  // the chunk's entries are contiguous in the workspace (list order)
      const float4* src = reinterpret_cast<const float4*>(
          a.ws_taps + ((long)(a.unit_base[li] + u) * a.R + cb) * TS);
      for (int i = tid; i < ncur * (TS / 4); i += THREADS) reinterpret_cast<float4*>(tab)[i] = src[i];
      return;
    }
    for (int i = tid; i < ncur * (TS / 2); i += THREADS) {
      const int j = i / (TS / 2), e2 = i - j * (TS / 2);
      const float2 v = *reinterpret_cast<const float2*>(cob + list[cb + j] * CW + 2 * e2);
      *reinterpret_cast<float2*>(tab + j * TS + 2 * e2) = v;
    }
  };

  // Fixed point needs a bound on max|dY| of the workgroup's items before the first add, but a
  // sweep over all of them up front costs a second pass of loads (measured: +10 us per launch).
  // So the scale is OPTIMISTIC: it is derived from the first chunk's items, which are in registers
  // anyway (2^30 of the 2^31 range is used, so the true maximum may be up to twice that without
  // any risk of overflow); every thread keeps the running maximum of what it actually streams, and
  // if the workgroup-wide maximum turns out larger (heavy-tailed gradients; never for the
  // near-Gaussian ones of the baseline), or a later item is non-finite, the band is zeroed and
  // accumulated again with the exact maximum (or with float adds).  The result is the same
  // deterministic function of the inputs either way.
  // maxima as bit patterns of |g| (absbits4: a NaN survives); anything above FLT_MAX's pattern is non-finite
  auto wave_max_to = [&](unsigned m, int slot) {
    m = wave_max_u32(m);
    if ((tid & (kWave - 1)) == 0) {
      atomicMax(reinterpret_cast<unsigned*>(nlist + slot), m);
      if (m > kFltMaxBits) atomicOr(nlist + 3, 1);
    }
  };
  auto set_scale = [&](float gmax) {
    const float bound = gmax * (float)nlist[1];  // no pixel of the band can exceed this
    // One unit is bound * 2^-30: with a weight bound above 2048 (hundreds of sub-pixel bins piled
    // onto one pixel of a tiny map -- 20x the baseline's bands) it would exceed max|dY| * 4e-6 and
    // the rounding of a few hundred adds could reach 1e-4: such a band takes the float adds.
    if (!(bound <= FLT_MAX) || nlist[1] > 2048) { use_fx = false; return; }
    int e;
    frexpf(bound, &e);  // bound < 2^e
    const int S = iminr(30 - e, 126);
    fx_scale = ldexpf(1.f, S);
    fx_inv = ldexpf(1.f, -S);
  };
  Item cur;
  const int tch = FLT ? (nl > 0 ? nl : 1) : TCH;  // no tables to stage: the whole list is one chunk
  load_item(tid, 0, iminr(tch, nl) * GP, cur);
  stage_tables(0, iminr(tch, nl));
  unsigned m_all = absbits4(cur.g);
  if (use_fx) wave_max_to(m_all, 2);
  __syncthreads();
  float gmax_used = 0.f;
  // margins / count of the sampled non-zero gradients this thread streams: the dynamic-range verdict behind the
  // scatter (kFxRangeBits, common.h)
  int e_acc = 0, e_thr = 0, e_mask = 3;
  if (use_fx) {
    gmax_used = __uint_as_float((unsigned)nlist[2]);
    if (nlist[3]) use_fx = false;  // non-finite gradients: float adds (a zeroed band is 0 in both formats)
    else set_scale(gmax_used > 0.f ? gmax_used : 1.f);
    e_thr = fx_range_thr(gmax_used, nlist[1]);
    e_mask = fx_range_stride_mask(THREADS, ((long)nl * GP + THREADS - 1) / THREADS, 3);   // (every fourth trip at the baseline)
  }
  bool synced = false;  // the scatter is already fenced by a barrier
  for (int attempt = 0; attempt < 2; ++attempt) {
    for (int cb = 0; cb < nl; cb += tch) {
      const int ncur = iminr(tch, nl - cb);
      const int nli = ncur * GP;  // lane items of this chunk
      if (cb > 0 || attempt > 0) {
        load_item(tid, cb, nli, cur);
        __syncthreads();  // the previous chunk's tables are no longer read
        stage_tables(cb, ncur);
        __syncthreads();
      }
      for (int t = tid; t < nli; t += THREADS) {
        Item nxt;
        load_item(t + THREADS, cb, nli, nxt);
        m_all = umaxr(m_all, absbits4(cur.g));
        if (((t / THREADS) & e_mask) == 0) e_acc += fx_range_sample(cur.g.x, e_thr);   // (wave uniform)
        if (TAPS) scatter_taps(cur, cur.j);
        else scatter_item(cur, cur.j);
        cur = nxt;
      }
    }
    if (!use_fx || attempt > 0) break;
    // was the optimistic scale enough, and is the unit fine enough for what was streamed?  (checked behind
    // the barrier that ends the scatter anyway)
    wave_max_to(m_all, 4);
    {
      const int es = wave_sum_i32(e_acc);
      if ((tid & (kWave - 1)) == 0) atomicAdd(nlist + 5, es);   // integer sums: the order of the waves does not matter
    }
    __syncthreads();
    const float gmax_true = __uint_as_float((unsigned)nlist[4]);
    const bool fine = fx_range_fine(nlist[5], gmax_used, gmax_true);
    if (!nlist[3] && fine && gmax_true <= 2.f * gmax_used) { synced = true; break; }  // also when all gradients are zero
    __syncthreads();  // every thread has read the verdict before the band is cleared
    // rare: accumulate the band again with the exact maximum, or -- non-finite gradients, a dynamic range the
    // fixed-point unit is too coarse for -- with float adds
    {
      float4* p4 = reinterpret_cast<float4*>(smem);
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      for (int i = tid; i < plane_pad / 4; i += THREADS) p4[i] = z;
    }
    if (nlist[3] || !fine) use_fx = false;
    else set_scale(gmax_true);
  }
  if (!synced) __syncthreads();
  if (SD_ABLATE(a, 2)) return;  // (profiling build, 2: no write-out)
  const long off = (((long)img * a.C + c) * H + row0) * W;
  TIO* dst = reinterpret_cast<TIO*>(a.dx[lvl]) + off;
  if (((off | band_elems) & 3) == 0) {
    for (int i = tid; i < band_elems / 4; i += THREADS) {
      float4 v = reinterpret_cast<const float4*>(plane)[i];
      if (use_fx) {
        const int4 q = reinterpret_cast<const int4*>(plane_i)[i];
        v = make_float4((float)q.x * fx_inv, (float)q.y * fx_inv, (float)q.z * fx_inv, (float)q.w * fx_inv);
      }
      if constexpr (HALF) {
        __half2* d2 = reinterpret_cast<__half2*>(dst) + 2 * i;   // (8-byte aligned: off and band_elems % 4 == 0)
        if (a.req == SD_REQ_ADD) {   // the sum is formed in fp32, rounded once
          const float2 o0 = __half22float2(d2[0]), o1 = __half22float2(d2[1]);
          v.x += o0.x; v.y += o0.y; v.z += o1.x; v.w += o1.y;
        }
        uint2 pk;
        const __half2 h0 = __floats2half2_rn(v.x, v.y), h1 = __floats2half2_rn(v.z, v.w);
        pk.x = *reinterpret_cast<const unsigned*>(&h0);
        pk.y = *reinterpret_cast<const unsigned*>(&h1);
        *reinterpret_cast<uint2*>(d2) = pk;
      } else {
        float4* d4 = reinterpret_cast<float4*>(dst);
        if (a.req == SD_REQ_ADD) {
          const float4 o = d4[i];
          v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
        }
        d4[i] = v;
      }
    }
  } else {
    for (int i = tid; i < band_elems; i += THREADS) {
      float v = use_fx ? (float)plane_i[i] * fx_inv : plane[i];
      if constexpr (HALF) {
        if (a.req == SD_REQ_ADD) v += __half2float(dst[i]);
        dst[i] = __float2half(v);
      } else {
        dst[i] = (a.req == SD_REQ_ADD) ? dst[i] + v : v;
      }
    }
  }
}

// The drop-in ROIAlign_v2 backward on a map whose four channel planes fit in LDS together (the C4
// family: (2,1024,50,84), `config/faster_r50v1c4_c5_512roi_1x.py:90-94`), round 3.  Per channel the
// out_grad / arg-max rows of a RoI are 196 bytes at a stride of C * 196: one channel per workgroup
// reads 196-byte fragments.  Here a workgroup owns FOUR consecutive channels of one image: the rows
// of (RoI, c..c+3) are 784 contiguous, 16-byte aligned bytes of each of the three inputs, read as
// dwordx4, and the four dX planes leave as one contiguous run.  The scatter is the reference's
// (roi_align_v2.cu:35-84: every bin with an arg-max adds its four bilinear terms).
// Round 5: the sums are 32-bit fixed point (integer LDS adds: 4-9 lane adds per clock and CU against ~2
// for the fp32 compare-and-swap loop, which was half of this kernel's time).  One unit is
// 2^-30 of (max|dY| x the plane's weight bound).  The weight bound is per PIXEL here: a RoI can put at
// most nx * ny bins' worth of weight on one pixel (the bins whose samples come within a pixel of it, as
// in bwd_band_list), and only on the pixels of its own extent -- so every RoI adds nx * ny to its
// rectangle of a 2-D difference array (four integer adds), two prefix passes turn that into the bound
// of every pixel, and its maximum is the plane's.  (The band kernels sum nx * ny over ALL RoIs of a band:
// 512 RoIs of 16 would always exceed their cap of 2048.)  max|dY| is taken optimistically from each
// thread's first item with a factor two of headroom and verified behind the scatter, as in
// roi_align_bwd_packed4; non-finite gradients, a weight bound above 2048 and `roi_align_bwd_fx` = 0
// take the fp32 compare-and-swap adds.  The result is a deterministic function of the inputs.
//   grid: x = channel quad, y = image; LDS = 4 * H * W words + 8 control words
template <int THREADS>
__global__ __launch_bounds__(THREADS) void roi_align_bwd_flt4_kernel(BwdFusedArgs a, int lvl) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int CC = 4;
  const int tid = threadIdx.x;
  const int G = a.C / CC;
  const int c = CC * ((G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x);
  const int img = blockIdx.y;
  const int H = a.L.H[lvl], W = a.L.W[lvl], HW = H * W, PP = a.PP;
  float* plane = smem;  // [CC][HW]
  int* plane_i = reinterpret_cast<int*>(smem);
  int* ctl = plane_i + CC * HW;  // [0] weight bound, [1] max|dY| of the first items, [2] non-finite flag, [3] true max|dY|, [4] range margins | sample count
  bool use_fx = !a.float_adds;
  if (tid < 8) ctl[tid] = 0;
  if (use_fx) {
    // per-pixel weight bound: difference array of (H + 1) x (W + 1) words in the (not yet used) planes
    const int DW = W + 1;
    for (int i = tid; i < (H + 1) * DW; i += THREADS) plane_i[i] = 0;
    __syncthreads();
    const int pw = PP == 49 ? 7 : 14;  // the launcher admits 7x7 and 14x14
    const float scale = a.L.scale[lvl];
    for (int r = tid; r < a.R; r += THREADS) {
      const float4 rb = *reinterpret_cast<const float4*>(a.rois + ((long)img * a.R + r) * 4);
      // pixels the taps of this RoI can reach: the clipped box +-2 (bwd_band_list); anything else
      // (NaN / inf coordinates) counts on the whole plane
      const float xs = fminr(fmaxr(rb.x * scale, 0.f), (float)(W - 1)), xe = fminr(fmaxr(rb.z * scale, 0.f), (float)(W - 1));
      const float ys = fminr(fmaxr(rb.y * scale, 0.f), (float)(H - 1)), ye = fminr(fmaxr(rb.w * scale, 0.f), (float)(H - 1));
      const float xlo = fminr(xs, xe) - 2.f, xhi = fmaxr(xs, xe) + 2.f;
      const float ylo = fminr(ys, ye) - 2.f, yhi = fmaxr(ys, ye) + 2.f;
      int x0 = 0, x1 = W - 1, y0 = 0, y1 = H - 1;
      if (xlo >= -2.f && xhi <= (float)(W + 1)) { x0 = imaxr((int)floorf(xlo), 0); x1 = iminr((int)ceilf(xhi), W - 1); }
      if (ylo >= -2.f && yhi <= (float)(H + 1)) { y0 = imaxr((int)floorf(ylo), 0); y1 = iminr((int)ceilf(yhi), H - 1); }
      const float bwx = (rb.z - rb.x) * scale * (1.f / (float)pw), bwy = (rb.w - rb.y) * scale * (1.f / (float)pw);
      const float fx = 2.00002f * __builtin_amdgcn_rcpf(bwx), fy = 2.00002f * __builtin_amdgcn_rcpf(bwy);
      const int nx = (bwx > 0.f && fx < (float)pw) ? iminr((int)fx + 2, pw) : pw;
      const int ny = (bwy > 0.f && fy < (float)pw) ? iminr((int)fy + 2, pw) : pw;
      const int w = nx * ny;
      atomicAdd(plane_i + y0 * DW + x0, w);
      atomicAdd(plane_i + y0 * DW + x1 + 1, -w);
      atomicAdd(plane_i + (y1 + 1) * DW + x0, -w);
      atomicAdd(plane_i + (y1 + 1) * DW + x1 + 1, w);
    }
    __syncthreads();
    for (int y = tid; y <= H; y += THREADS) {  // prefix along the rows (running sum in a register)
      int run = 0;
#pragma unroll 4
      for (int x = 0; x <= W; ++x) {
        run += plane_i[y * DW + x];
        plane_i[y * DW + x] = run;
      }
    }
    __syncthreads();
    int m = 0;
    for (int x = tid; x <= W; x += THREADS) {  // prefix down the columns: only its maximum is kept
      int run = 0;
#pragma unroll 4
      for (int y = 0; y <= H; ++y) {
        run += plane_i[y * DW + x];
        m = imaxr(m, run);
      }
    }
    if (m > 0) atomicMax(ctl, m);
    __syncthreads();
  }
  for (int i = tid; i < CC * HW; i += THREADS) plane_i[i] = 0;
  const int per = CC * PP / 4;  // float4 units per RoI (CC * PP is a multiple of 4)
  const int nunits = a.R * per;
  const long base = ((long)img * a.R * a.C + c) * PP;
  const long roi_stride = (long)a.C * PP;
  struct Item { float4 g, x, y; int L; };
  auto load_item = [&](int u, Item& it) {
    it.L = -1;
    if (u < nunits) {
      const int r = u / per, L = u - r * per;
      const long idx = base + r * roi_stride + 4 * L;
      it.g = *reinterpret_cast<const float4*>(a.dy + idx);
      it.x = *reinterpret_cast<const float4*>(a.ax + idx);
      it.y = *reinterpret_cast<const float4*>(a.ay + idx);
      it.L = L;
    }
  };
  // maxima as bit patterns of |g| (absbits4: a NaN survives); anything above FLT_MAX's pattern is non-finite
  auto wave_max_to = [&](unsigned m, int slot) {
    m = wave_max_u32(m);
    if ((tid & (kWave - 1)) == 0) {
      atomicMax(reinterpret_cast<unsigned*>(ctl + slot), m);
      if (m > kFltMaxBits) atomicOr(ctl + 2, 1);
    }
  };
  float fx_scale = 1.f, fx_inv = 1.f;
  auto set_scale = [&](float gmax) {
    const float bound = gmax * (float)ctl[0];  // no pixel of a plane can exceed this
    if (!(bound <= FLT_MAX) || ctl[0] > 2048) { use_fx = false; return; }
    int e;
    frexpf(bound > 0.f ? bound : 1.f, &e);  // bound < 2^e
    const int S = iminr(30 - e, 126);
    fx_scale = ldexpf(1.f, S);
    fx_inv = ldexpf(1.f, -S);
  };
  Item cur;
  load_item(tid, cur);
  unsigned m_all = cur.L >= 0 ? absbits4(cur.g) : 0u;
  if (use_fx) wave_max_to(m_all, 1);
  __syncthreads();  // the planes are zero, the first maxima are in
  float gmax_used = 0.f;
  int e_acc = 0, e_thr = 0, e_mask = 7;  // margins / count of the sampled non-zero gradients (kFxRangeBits, common.h)
  if (use_fx) {
    gmax_used = __uint_as_float((unsigned)ctl[1]);
    // (the non-finite flag has its own barrier-separated read: a wave that reaches the verdict of attempt 0
    // early ORs into ctl[2] only after every wave has passed the barrier above)
    if (ctl[2]) use_fx = false;
    else set_scale(gmax_used > 0.f ? gmax_used : 1.f);
    e_thr = fx_range_thr(gmax_used, ctl[0]);
    e_mask = fx_range_stride_mask(THREADS, ((long)nunits + THREADS - 1) / THREADS, 7);
  }
  __syncthreads();  // every wave has read ctl[1] / ctl[2] before a fast wave's verdict can change them
  for (int attempt = 0; attempt < 2; ++attempt) {
    if (attempt > 0) load_item(tid, cur);
    for (int u = tid; u < nunits; u += THREADS) {
      Item nxt;
      load_item(u + THREADS, nxt);
      m_all = umaxr(m_all, absbits4(cur.g));
      if (((u / THREADS) & e_mask) == 0) e_acc += fx_range_sample(cur.g.x, e_thr);   // (wave uniform)
      const float gg[4] = {cur.g.x, cur.g.y, cur.g.z, cur.g.w}, xx[4] = {cur.x.x, cur.x.y, cur.x.z, cur.x.w};
      const float yy[4] = {cur.y.x, cur.y.y, cur.y.z, cur.y.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float a_x = xx[k], a_y = yy[k];
        if (a_x == -1.f || a_y == -1.f) continue;  // roi_align_v2.cu:53: nothing was pooled
        const int e = 4 * cur.L + k;
        const int cc = (e >= PP) + (e >= 2 * PP) + (e >= 3 * PP);
        const int hlow = iminr(imaxr((int)floorf(a_y), 0), H - 1);
        const int hhigh = iminr(imaxr((int)ceilf(a_y), 0), H - 1);
        const int wleft = iminr(imaxr((int)floorf(a_x), 0), W - 1);
        const int wright = iminr(imaxr((int)ceilf(a_x), 0), W - 1);
        // (v - low) / (high - low) with high - low == 1
        const float alpha = (hlow == hhigh) ? 0.5f : (a_y - (float)hlow);
        const float beta = (wleft == wright) ? 0.5f : (a_x - (float)wleft);
        const float g = gg[k];
        const float w00 = g * (1 - alpha) * (1 - beta), w01 = g * (1 - alpha) * beta;
        const float w10 = g * alpha * (1 - beta), w11 = g * alpha * beta;
        const int o0 = cc * HW + hlow * W, o1 = cc * HW + hhigh * W;
        if (use_fx) {
          lds_add_i32(plane_i + o0 + wleft, w00, fx_scale);
          lds_add_i32(plane_i + o0 + wright, w01, fx_scale);
          lds_add_i32(plane_i + o1 + wleft, w10, fx_scale);
          lds_add_i32(plane_i + o1 + wright, w11, fx_scale);
        } else {
          lds_add_cas(plane + o0 + wleft, w00);
          lds_add_cas(plane + o0 + wright, w01);
          lds_add_cas(plane + o1 + wleft, w10);
          lds_add_cas(plane + o1 + wright, w11);
        }
      }
      cur = nxt;
    }
    if (!use_fx || attempt > 0) break;
    // was the optimistic scale enough, and is the unit fine enough for what was streamed?
    wave_max_to(m_all, 3);
    {
      const int es = wave_sum_i32(e_acc);
      if ((tid & (kWave - 1)) == 0) atomicAdd(ctl + 4, es);   // integer sums: the order of the waves does not matter
    }
    __syncthreads();
    const float gmax_true = __uint_as_float((unsigned)ctl[3]);
    const bool fine = fx_range_fine(ctl[4], gmax_used, gmax_true);
    if (!ctl[2] && fine && gmax_true <= 2.f * gmax_used) break;  // also when all gradients are zero
    __syncthreads();  // every thread has read the verdict before the planes are cleared
    // rare: again, with the exact maximum -- or with float adds (non-finite gradients, a dynamic range the
    // fixed-point unit is too coarse for)
    for (int i = tid; i < CC * HW; i += THREADS) plane_i[i] = 0;
    if (ctl[2] || !fine) use_fx = false;
    else set_scale(gmax_true);
    __syncthreads();
  }
  __syncthreads();
  float* dst = a.dx[lvl] + ((long)img * a.C + c) * HW;  // the four planes are contiguous
  for (int i = tid; i < CC * HW; i += THREADS) {
    const float v = use_fx ? (float)plane_i[i] * fx_inv : plane[i];
    dst[i] = (a.req == SD_REQ_ADD) ? dst[i] + v : v;
  }
}

// levels: dx[l] / H / W / scale from a.L; returns SD_ERR_UNSUPPORTED when a level does not fit
// prepass: 0 the launch builds its own lists / tap tables (roi_align_bwd_lists) when it has a workspace;
//          1 PLAN ONLY: fill `a` (bands, order, workspace pointers) for a list pre-pass somebody else
//            launches -- the forward's merged pre-pass -- and launch nothing; SD_ERR_UNSUPPORTED when
//            this shape / workspace does not get the list + tap-table form;
//          2 the lists and tap tables in the workspace are already built (by a prepass = 1 plan
//            of the same shapes, knobs and workspace): skip the pre-pass launch.
int launch_bwd_fused(BwdFusedArgs& a, int nlvl, hipStream_t st, void* workspace, size_t workspace_bytes, int prepass) {
  // packed arg-max: the wide-load kernel (roi_align_bwd_packed4); its coordinate tables share the
  // LDS with the band, so the band budget is a little smaller
  const bool flt = !a.amax8 && a.ax && a.ay;  // float arg-max planes: the same kernel without tables
  SD_REQUIRE(a.amax8 || flt, "RoIAlign backward needs the forward's arg-max (packed bytes or the two float planes)");
  a.float_adds = tuning("roi_align_bwd_fx", 1) == 0 ? 1 : 0;
  const int tch = a.PP == 49 ? 32 : 16;   // RoIs per staged coordinate-table chunk
  const int ne = a.PP == 49 ? 3 * 14 : 3 * 28;  // sample coordinates per RoI
  a.ablate = SD_PROF_TUNING("roi_align_bwd_ablate", 0);
  if ((long)a.R * a.C * a.PP >= (1L << 31)) return SD_ERR_UNSUPPORTED;  // 32-bit lane offsets
  // single-level float arg-max backward on a small map: four channels per workgroup
  if (prepass == 0 && flt && !a.filter && nlvl == 1 && a.dx[0] && a.C % 4 == 0 && a.B <= 65535 &&
      (long)a.L.H[0] * a.L.W[0] * 16 <= 72 * 1024 && tuning("roi_align_bwd_flt4", 1) == 1 &&
      (((uintptr_t)a.dy | (uintptr_t)a.ax | (uintptr_t)a.ay) & 15) == 0) {
    const size_t lds4 = (size_t)a.L.H[0] * a.L.W[0] * 16 + 32;
    if (lds4 > 64 * 1024)
      SD_HIP_CHECK(hipFuncSetAttribute((const void*)roi_align_bwd_flt4_kernel<512>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
    hipLaunchKernelGGL((roi_align_bwd_flt4_kernel<512>), dim3(a.C / 4, a.B), dim3(512), lds4, st, a, 0);
    note_dispatch("sd::roi_align_bwd_flt4_kernel<512>");
    SD_LAUNCH_CHECK();
    return SD_OK;
  }
  size_t lds_max = 0;
  long work[SD_MAX_FPN_LEVELS];
  int nl = 0;
  long units = 0;
  // bands of every level for a band budget and a table entry size (words per sample coordinate)
  auto plan = [&](long budget, int entry_words) -> int {
    const size_t tab_bytes = !flt ? (size_t)tch * ne * entry_words * 4 : 0;
    lds_max = 0;
    nl = 0;
    units = 0;
    for (int l = 0; l < nlvl; ++l) {
      if (!a.dx[l]) continue;
      if (a.L.H[l] > 32767 || a.L.W[l] > 32767) return SD_ERR_UNSUPPORTED;  // packed neighbour pairs
      const long plane_bytes = (long)a.L.H[l] * a.L.W[l] * 4;
      int nb = (int)((plane_bytes + budget - 1) / budget);
      if (nb < 1) nb = 1;
      int rows = (a.L.H[l] + nb - 1) / nb;
      nb = (a.L.H[l] + rows - 1) / rows;
      a.band_rows[l] = rows;
      a.nbands[l] = nb;
      if ((long)rows * a.L.W[l] >= 65535) return SD_ERR_UNSUPPORTED;  // 16-bit band offsets
      const size_t lds =
          (size_t)((((long)rows * a.L.W[l] + 3) & ~3L) * 4) + tab_bytes + (size_t)(a.R + 8 + 16) * 4;
      if (lds > 150 * 1024) return SD_ERR_UNSUPPORTED;
      if (lds > lds_max) lds_max = lds;
      work[l] = (long)a.B * nb * a.C;
      units += (long)a.B * nb;
      a.order[nl++] = l;
    }
    return SD_OK;
  };
  // With the workspace pre-pass providing 8-byte tap entries the tables take twice the LDS; the band
  // budget drops from 36 to 27 KB so that FOUR workgroups still share a CU (27 + 10.75 + 2.1 KB):
  // 86.6 us against 98.5 us with 36 KB bands at three per CU.
  const int lists_mode = tuning("roi_align_bwd_lists", 1);  // 1 lists + taps, 2 lists only, 0 none
  bool use_taps = false, use_lists = false;
  size_t list_bytes = 0;
  if (!flt && workspace && ((uintptr_t)workspace & 15) == 0 && lists_mode == 1) {
    if (int e = plan(27L * 1024, 2)) return e;
    list_bytes = (((size_t)units * (a.R + 2) * sizeof(int)) + 15) & ~(size_t)15;
    use_taps = workspace_bytes >= list_bytes + (size_t)units * a.R * 2 * ne * sizeof(float);
    // (the tap entries hold 16-bit BYTE offsets into the band: a band wider than 64 KB -- a single row of > 16 K
    // pixels -- keeps the coordinate-table form)
    for (int l = 0; l < nlvl; ++l)
      if (a.dx[l] && (long)a.band_rows[l] * a.L.W[l] * 4 >= 65535) use_taps = false;
    use_lists = use_taps;
  }
  if (!use_taps) {
    if (int e = plan(36L * 1024, 1)) return e;
    list_bytes = (((size_t)units * (a.R + 2) * sizeof(int)) + 15) & ~(size_t)15;
    use_lists = workspace && lists_mode >= 1 && workspace_bytes >= list_bytes;
  }
  // Launch order = expected duration of ONE workgroup, longest first: a level that fits in one
  // band sees all of its image's RoIs in every workgroup (2x the items of a P2 band at the
  // baseline), so the few-band levels go first and the many short P2 bands fill the tail.
  for (int i = 0; i < nl; ++i)
    for (int j = i + 1; j < nl; ++j)
      if (a.nbands[a.order[j]] < a.nbands[a.order[i]] ||
          (a.nbands[a.order[j]] == a.nbands[a.order[i]] &&
           (long)a.L.H[a.order[j]] * a.L.W[a.order[j]] > (long)a.L.H[a.order[i]] * a.L.W[a.order[i]])) {
        const int t = a.order[i];
        a.order[i] = a.order[j];
        a.order[j] = t;
      }
  long total = 0;
  for (int i = 0; i < nl; ++i) {
    total += work[a.order[i]];
    a.block_end[i] = (int)total;
  }
  a.nlaunch = nl;
  if (total == 0) return SD_OK;
  if (total >= (1L << 31)) return SD_ERR_UNSUPPORTED;
  // RoI lists (and tap tables) of the (level, image, band) units, once per launch instead of once
  // per channel
  a.ws_list = nullptr;
  a.ws_taps = nullptr;
  {
    long ub = 0;
    for (int i = 0; i < nl; ++i) {
      a.unit_base[i] = (int)ub;
      ub += (long)a.B * a.nbands[a.order[i]];
    }
    if (nl < SD_MAX_FPN_LEVELS) a.unit_base[nl] = (int)ub;
  }
  a.lists_units = 0;
  a.pix_bound_words = 0;
  if (tuning("roi_align_bwd_pixbound", 1) == 1)
    for (int i = 0; i < nl; ++i) {
      const int l = a.order[i];
      const long words = (long)(((a.band_rows[l] - 1) >> 2) + 2) * (((a.L.W[l] - 1) >> 2) + 2);   // 4 x 4 cells (bwd_lists_block)
      if (words <= kPixBoundMaxWords && words > a.pix_bound_words) a.pix_bound_words = (int)words;
    }
  if (use_lists && nl < SD_MAX_FPN_LEVELS) {
    a.ws_list = static_cast<int*>(workspace);
    if (use_taps) a.ws_taps = reinterpret_cast<float*>(static_cast<char*>(workspace) + list_bytes);
    a.lists_units = (int)units;
    if (prepass == 0)
      if (int e = launch_bwd_lists(a, (int)units, st)) return e;
  }
  if (prepass != 0 && !(a.ws_list && a.ws_taps)) return SD_ERR_UNSUPPORTED;
  if (prepass == 1) return SD_OK;
  // the wide-load kernel: 512 lanes, coordinate tables staged 32 (7x7) / 16 (14x14) RoIs at a time
  // MODE 0: packed arg-max, tables derived per workgroup; 1: tap tables from the list pre-pass; 2: float arg-max planes
#define SD_BWDW(PHv, TCHv, HALFv)                                                                \
  do {                                                                                           \
    auto k = flt ? roi_align_bwd_packed4<PHv, PHv, 512, TCHv, 2, false>                          \
                 : a.ws_taps ? roi_align_bwd_packed4<PHv, PHv, 512, TCHv, 1, HALFv>              \
                             : roi_align_bwd_packed4<PHv, PHv, 512, TCHv, 0, HALFv>;             \
    if (lds_max > 64 * 1024)                                                                     \
      SD_HIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                       (int)lds_max));                                           \
    hipLaunchKernelGGL(k, dim3((unsigned)total), dim3(512), lds_max, st, a);                     \
  } while (0)
  if (a.half_io) {   // fp16 I/O: packed arg-max only
    if (flt) return SD_ERR_UNSUPPORTED;
    if (a.PP == 49) SD_BWDW(7, 32, true); else SD_BWDW(14, 16, true);
  } else {
    if (a.PP == 49) SD_BWDW(7, 32, false); else SD_BWDW(14, 16, false);
  }
#undef SD_BWDW
  note_dispatch("%ssd::roi_align_bwd_packed4<%d,%d,512,%d,%d%s>", prepass == 0 && a.ws_list ? "sd::roi_align_bwd_lists + " : "",
                a.PP == 49 ? 7 : 14, a.PP == 49 ? 7 : 14, tch, flt ? 2 : (a.ws_taps ? 1 : 0), a.half_io ? ",true" : "");
  SD_LAUNCH_CHECK();
  return SD_OK;
}

template <int PP>
static int launch_bwd_plane(BwdArgs& a, hipStream_t st) {
  // LDS budget per workgroup: a plane larger than it is cut into row bands; several small planes
  // (CPB channels) share a workgroup only while that still leaves >= 1024 workgroups
  constexpr bool fx = true;   // int64 fixed-point planes (bit-reproducible sums; float CAS inside on non-finite dY)
  const long esz = fx ? 8 : 4;
  const long plane_bytes = (long)a.H * a.W * esz;
  const long budget = 72L * 1024;
  int cpb = 1;
  a.nbands = 1;
  a.band_rows = a.H;
  if (plane_bytes <= budget) {
    for (int c : {8, 4, 2})
      if (c * plane_bytes <= budget && a.C % c == 0 && (long)a.B * a.C / c >= 1024) {
        cpb = c;
        break;
      }
  } else {
    a.nbands = (int)((plane_bytes + budget - 1) / budget);
    a.band_rows = (a.H + a.nbands - 1) / a.nbands;
    a.nbands = (a.H + a.band_rows - 1) / a.band_rows;
  }
  const long band_elems = (long)a.band_rows * a.W;
  const size_t lds = (size_t)(((cpb * band_elems + 3) & ~3L) * esz) + (size_t)(a.R + 4) * 4;
  SD_REQUIRE(lds <= 160 * 1024, "RoIAlign backward needs %zu B of LDS (W=%d R=%d too large)", lds,
             a.W, a.R);
  const int grid = a.B * a.nbands * (a.C / cpb);
  const int threads = lds > 96 * 1024 ? 1024 : (lds > 24 * 1024 ? 512 : 256);
#define SD_BWD_LAUNCH(CPB, T, FX)                                                               \
  do {                                                                                          \
    auto k = roi_align_bwd_plane<PP, CPB, T, FX>;                                               \
    if (lds > 64 * 1024)                                                                        \
      SD_HIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                       (int)lds));                                              \
    hipLaunchKernelGGL(k, dim3(grid), dim3(T), lds, st, a);                                     \
  } while (0)
#define SD_BWD_T(CPB, FX)                                  \
  do {                                                     \
    if (threads == 1024) SD_BWD_LAUNCH(CPB, 1024, FX);     \
    else if (threads == 512) SD_BWD_LAUNCH(CPB, 512, FX);  \
    else SD_BWD_LAUNCH(CPB, 256, FX);                      \
  } while (0)
#define SD_BWD_C(FX)                   \
  do {                                 \
    if (cpb == 1) SD_BWD_T(1, FX);     \
    else if (cpb == 2) SD_BWD_T(2, FX); \
    else if (cpb == 4) SD_BWD_T(4, FX); \
    else SD_BWD_T(8, FX);              \
  } while (0)
  SD_BWD_C(true);
#undef SD_BWD_C
#undef SD_BWD_T
#undef SD_BWD_LAUNCH
  SD_LAUNCH_CHECK();
  return SD_OK;
}

int launch_bwd(BwdArgs& a, hipStream_t st) {
  const long count = (long)a.B * a.R * a.C * a.PP;
  const size_t dx_bytes = (size_t)a.B * a.C * a.H * a.W * 4;
  if (dx_bytes == 0) return SD_OK;
  const int variant = tuning("roi_align_bwd", 1);  // 0 global atomics, 1 LDS planes
  a.ablate = SD_PROF_TUNING("roi_align_bwd_ablate", 0);
  const size_t list_bytes = (size_t)(a.R + 8) * 4;
  if (variant >= 1 && (a.PP == 49 || a.PP == 196) && list_bytes < 20 * 1024 && count > 0) {
    return a.PP == 49 ? launch_bwd_plane<49>(a, st) : launch_bwd_plane<196>(a, st);
  }
  if (a.req == SD_REQ_WRITE) SD_HIP_CHECK(hipMemsetAsync(a.dx, 0, dx_bytes, st));
  if (count == 0) return SD_OK;
  const int grid = (int)((count + 255) / 256 < 65536 * 16 ? (count + 255) / 256 : 65536 * 16);
  hipLaunchKernelGGL(roi_align_bwd_atomic, dim3(grid), dim3(256), 0, st, a);
  SD_LAUNCH_CHECK();
  return SD_OK;
}

}  // namespace sd

using namespace sd;

extern "C" int sd_roi_align_v2_bwd(const float* out_grad, const float* rois, const float* maxidx_x,
                                   const float* maxidx_y, float* d_data, float* d_rois,
                                   int req_data, int req_rois, int B, int C, int H, int W, int R,
                                   int pooled_h, int pooled_w, float spatial_scale, void* stream) {
  if (int e = check_dims(B, C, R, pooled_h, pooled_w)) return e;
  SD_REQUIRE(H > 0 && W > 0 && (long)H * W < (1L << 30), "bad feature size %d x %d", H, W);
  SD_REQUIRE(req_data == SD_REQ_NULL || req_data == SD_REQ_WRITE || req_data == SD_REQ_ADD,
             "ROIAlign: Backward doesn't support req_data=%d (kWriteInplace)", req_data);
  SD_REQUIRE(req_rois == SD_REQ_NULL || req_rois == SD_REQ_WRITE || req_rois == SD_REQ_ADD,
             "ROIAlign: Backward doesn't support req_rois=%d (kWriteInplace)", req_rois);
  hipStream_t st = (hipStream_t)stream;
  if (req_data != SD_REQ_NULL) {
    SD_REQUIRE(d_data && ((out_grad && maxidx_x && maxidx_y && rois) || (long)B * R * C == 0),
               "null tensor pointer");
    BwdArgs a{};
    a.L.nlvl = 1;
    a.dy = out_grad; a.ax = maxidx_x; a.ay = maxidx_y; a.rois = rois; a.dx = d_data;
    a.B = B; a.C = C; a.R = R; a.PP = pooled_h * pooled_w; a.H = H; a.W = W;
    a.scale = spatial_scale;
    a.filter_lvl = -1;
    a.req = req_data;
    bool done = false;
    // (the fused kernels' weight bounds count bins per axis: square 7x7 / 14x14 pools only -- a 1x49 pool has
    // PP = 49 too and goes to the plane kernels, whose 64-bit sums need no such bound)
    if (tuning("roi_align_bwd", 2) == 2 && (a.PP == 49 || a.PP == 196) && pooled_h == pooled_w &&
        (long)B * R * C > 0 && R <= 8192) {
      BwdFusedArgs f{};
      f.L = a.L;
      f.L.nlvl = 1; f.L.H[0] = H; f.L.W[0] = W; f.L.scale[0] = spatial_scale;
      f.dy = out_grad; f.ax = maxidx_x; f.ay = maxidx_y; f.rois = rois;
      f.dx[0] = d_data;
      f.B = B; f.C = C; f.R = R; f.PP = a.PP; f.filter = 0; f.req = req_data;
      const int e = launch_bwd_fused(f, 1, st);
      if (e != SD_ERR_UNSUPPORTED) {
        if (e) return e;
        done = true;
      }
    }
    if (!done)
      if (int e = launch_bwd(a, st)) return e;
  }
  if (req_rois == SD_REQ_WRITE && (long)B * R > 0) {  // roi_align_v2.cu:139-141
    SD_REQUIRE(d_rois, "d_rois is null but req_rois == write");
    SD_HIP_CHECK(hipMemsetAsync(d_rois, 0, (size_t)B * R * 4 * sizeof(float), st));
  }
  return SD_OK;
}

extern "C" size_t sd_fpn_roi_align_bwd_workspace_bytes(const int* Hs_host, const int* Ws_host, int nlvl,
                                                       int B, int R) {
  if (!Hs_host || !Ws_host || nlvl <= 0 || B <= 0 || R <= 0) return 0;
  const long budget = 27 * 1024;  // the smallest band the launcher uses gives the most units
  long units = 0;
  for (int l = 0; l < nlvl; ++l) {
    const long plane_bytes = (long)Hs_host[l] * Ws_host[l] * 4;
    long nb = (plane_bytes + budget - 1) / budget;
    if (nb < 1) nb = 1;
    units += (long)B * (nb + 1);
  }
  // lists + the tap tables of the larger pooled size (14x14: 2 * 3 * 28 words per RoI)
  return ((((size_t)units * (R + 2) * sizeof(int)) + 15) & ~(size_t)15) + (size_t)units * R * 168 * sizeof(float) + 16;
}

extern "C" int sd_fpn_roi_align_bwd_packed(const float* out_grad, const float* rois,
                                           const uint8_t* argmax, const float* coords,
                                           float* const* d_feats_host,
                                           const int* Hs_host, const int* Ws_host,
                                           const int* strides_host, int nlvl, int req_data, int B,
                                           int C, int R, int pooled_h, int pooled_w,
                                           float roi_canonical_scale, float roi_canonical_level,
                                           void* stream) {
  return sd_fpn_roi_align_bwd_packed_ws(out_grad, rois, argmax, coords, d_feats_host, Hs_host, Ws_host,
                                        strides_host, nlvl, req_data, B, C, R, pooled_h, pooled_w,
                                        roi_canonical_scale, roi_canonical_level, nullptr, 0, stream);
}

// mode: bit 0 the forward left the lists / tap tables in `workspace` (planned), bit 1 fp16 gradient in and out
static int fpn_bwd_packed_impl(int mode, const float* out_grad, const float* rois,
                                              const uint8_t* argmax, const float* coords,
                                              float* const* d_feats_host,
                                              const int* Hs_host, const int* Ws_host,
                                              const int* strides_host, int nlvl, int req_data, int B,
                                              int C, int R, int pooled_h, int pooled_w,
                                              float roi_canonical_scale, float roi_canonical_level,
                                              void* workspace, size_t workspace_bytes, void* stream) {
  if (int e = check_dims(B, C, R, pooled_h, pooled_w)) return e;
  SD_REQUIRE(d_feats_host && Hs_host && Ws_host && strides_host, "null level description");
  SD_REQUIRE(req_data == SD_REQ_NULL || req_data == SD_REQ_WRITE || req_data == SD_REQ_ADD,
             "ROIAlign: Backward doesn't support req_data=%d (kWriteInplace)", req_data);
  if (req_data == SD_REQ_NULL) return SD_OK;
  const int PPv = pooled_h * pooled_w;
  SD_REQUIRE((pooled_h == 7 && pooled_w == 7) || (pooled_h == 14 && pooled_w == 14),
             "packed arg-max backward supports 7x7 and 14x14 pooling only");
  SD_REQUIRE(R <= 8192, "packed arg-max backward: R=%d > 8192", R);
  BwdFusedArgs f{};
  if (int e = fill_levels(f.L, nullptr, Hs_host, Ws_host, strides_host, nlvl, roi_canonical_scale,
                          roi_canonical_level))
    return e;
  if (nlvl == 1) f.L.nlvl = 2, f.L.stride[1] = -1;
  f.dy = out_grad; f.amax8 = argmax; f.coords = coords; f.rois = rois;
  for (int l = 0; l < nlvl; ++l) {
    SD_REQUIRE(d_feats_host[l] || (long)B * C == 0, "d_feats[%d] null", l);
    f.dx[l] = d_feats_host[l];
  }
  const int planned = mode & 1;
  f.half_io = (mode >> 1) & 1;
  f.B = B; f.C = C; f.R = R; f.PP = PPv; f.filter = 1; f.req = req_data;
  if ((long)B * R * C == 0) {
    for (int l = 0; l < nlvl; ++l)
      if (req_data == SD_REQ_WRITE && d_feats_host[l])
        SD_HIP_CHECK(hipMemsetAsync(d_feats_host[l], 0,
                                    (f.half_io ? 2 : sizeof(float)) * (size_t)B * C * Hs_host[l] * Ws_host[l],
                                    (hipStream_t)stream));
    return SD_OK;
  }
  SD_REQUIRE(out_grad && rois && argmax && coords, "null tensor pointer");
  SD_REQUIRE(((uintptr_t)argmax & 3) == 0 && ((uintptr_t)coords & 7) == 0,
             "argmax must be 4-byte and coords 8-byte aligned");
  SD_REQUIRE(!workspace || ((uintptr_t)workspace & 3) == 0, "workspace must be 4-byte aligned");
  int e = SD_ERR_UNSUPPORTED;
  // planned: the forward (sd_fpn_roi_align_fwd_packed_plan) has left the band lists / tap tables in
  // `workspace`; where its plan did not apply (same deterministic decision here) the normal path runs
  if (planned) e = launch_bwd_fused(f, nlvl, (hipStream_t)stream, workspace, workspace_bytes, 2);
  if (e == SD_ERR_UNSUPPORTED) e = launch_bwd_fused(f, nlvl, (hipStream_t)stream, workspace, workspace_bytes, 0);
  if (e == SD_ERR_UNSUPPORTED)
    return fail(e, f.half_io ? "fp16 packed arg-max backward runs on the default wide kernel only (a level does not "
                               "fit LDS, or roi_align_bwd_packed / _threads / _tch were changed)"
                             : "packed arg-max backward: a level does not fit LDS");
  return e;
}

extern "C" int sd_fpn_roi_align_bwd_packed_ws(const float* out_grad, const float* rois,
                                              const uint8_t* argmax, const float* coords,
                                              float* const* d_feats_host,
                                              const int* Hs_host, const int* Ws_host,
                                              const int* strides_host, int nlvl, int req_data, int B,
                                              int C, int R, int pooled_h, int pooled_w,
                                              float roi_canonical_scale, float roi_canonical_level,
                                              void* workspace, size_t workspace_bytes, void* stream) {
  return fpn_bwd_packed_impl(0, out_grad, rois, argmax, coords, d_feats_host, Hs_host, Ws_host, strides_host,
                             nlvl, req_data, B, C, R, pooled_h, pooled_w, roi_canonical_scale,
                             roi_canonical_level, workspace, workspace_bytes, stream);
}

extern "C" int sd_fpn_roi_align_bwd_packed_f16(const void* out_grad, const float* rois,
                                               const uint8_t* argmax, const float* coords,
                                               void* const* d_feats_host, const int* Hs_host,
                                               const int* Ws_host, const int* strides_host, int nlvl,
                                               int req_data, int B, int C, int R, int pooled_h,
                                               int pooled_w, float roi_canonical_scale,
                                               float roi_canonical_level, void* workspace,
                                               size_t workspace_bytes, void* stream) {
  SD_REQUIRE(((uintptr_t)out_grad & 1) == 0, "out_grad must be 2-byte aligned");
  if (d_feats_host)
    for (int l = 0; l < nlvl; ++l)
      SD_REQUIRE(((uintptr_t)d_feats_host[l] & 7) == 0, "d_feats[%d] must be 8-byte aligned", l);
  return fpn_bwd_packed_impl(2, reinterpret_cast<const float*>(out_grad), rois, argmax, coords,
                             reinterpret_cast<float* const*>(d_feats_host), Hs_host, Ws_host, strides_host,
                             nlvl, req_data, B, C, R, pooled_h, pooled_w, roi_canonical_scale,
                             roi_canonical_level, workspace, workspace_bytes, stream);
}

extern "C" size_t sd_fpn_roi_align_plan_bytes(const int* Hs_host, const int* Ws_host, int nlvl, int B, int R) {
  return sd_fpn_roi_align_bwd_workspace_bytes(Hs_host, Ws_host, nlvl, B, R);
}

extern "C" int sd_fpn_roi_align_bwd_packed_plan(const float* out_grad, const float* rois,
                                                const uint8_t* argmax, const float* coords,
                                                float* const* d_feats_host, const int* Hs_host,
                                                const int* Ws_host, const int* strides_host, int nlvl,
                                                int req_data, int B, int C, int R, int pooled_h,
                                                int pooled_w, float roi_canonical_scale,
                                                float roi_canonical_level, const void* plan,
                                                size_t plan_bytes, void* stream) {
  SD_REQUIRE(plan && ((uintptr_t)plan & 15) == 0, "plan must be the 16-byte aligned buffer the forward filled");
  return fpn_bwd_packed_impl(1, out_grad, rois, argmax, coords, d_feats_host, Hs_host, Ws_host, strides_host,
                             nlvl, req_data, B, C, R, pooled_h, pooled_w, roi_canonical_scale,
                             roi_canonical_level, const_cast<void*>(plan), plan_bytes, stream);
}

extern "C" int sd_fpn_roi_align_bwd(const float* out_grad, const float* rois,
                                    const float* maxidx_x, const float* maxidx_y,
                                    float* const* d_feats_host, const int* Hs_host,
                                    const int* Ws_host, const int* strides_host, int nlvl,
                                    int req_data, int B, int C, int R, int pooled_h, int pooled_w,
                                    float roi_canonical_scale, float roi_canonical_level,
                                    void* stream) {
  if (int e = check_dims(B, C, R, pooled_h, pooled_w)) return e;
  SD_REQUIRE(d_feats_host && Hs_host && Ws_host && strides_host, "null level description");
  SD_REQUIRE(req_data == SD_REQ_NULL || req_data == SD_REQ_WRITE || req_data == SD_REQ_ADD,
             "ROIAlign: Backward doesn't support req_data=%d (kWriteInplace)", req_data);
  if (req_data == SD_REQ_NULL) return SD_OK;
  BwdArgs a{};
  if (int e = fill_levels(a.L, nullptr, Hs_host, Ws_host, strides_host, nlvl, roi_canonical_scale,
                          roi_canonical_level))
    return e;
  if (nlvl == 1) a.L.nlvl = 2, a.L.stride[1] = -1;
  a.dy = out_grad; a.ax = maxidx_x; a.ay = maxidx_y; a.rois = rois;
  a.B = B; a.C = C; a.R = R; a.PP = pooled_h * pooled_w;
  a.req = req_data;
  for (int l = 0; l < nlvl; ++l)
    SD_REQUIRE(d_feats_host[l] || (long)B * C == 0, "d_feats[%d] null", l);
  const long count = (long)B * R * C * a.PP;
  if (tuning("roi_align_bwd", 2) == 2 && (a.PP == 49 || a.PP == 196) && pooled_h == pooled_w && count > 0 &&
      R <= 8192) {
    BwdFusedArgs f{};
    f.L = a.L;
    f.dy = out_grad; f.ax = maxidx_x; f.ay = maxidx_y; f.rois = rois;
    for (int l = 0; l < nlvl; ++l) {
      f.L.H[l] = Hs_host[l];
      f.L.W[l] = Ws_host[l];
      f.dx[l] = d_feats_host[l];
    }
    f.B = B; f.C = C; f.R = R; f.PP = a.PP; f.filter = 1; f.req = req_data;
    const int e = launch_bwd_fused(f, nlvl, (hipStream_t)stream);
    if (e != SD_ERR_UNSUPPORTED) return e;
  }
  for (int l = 0; l < nlvl; ++l) {
    a.dx = d_feats_host[l];
    a.H = Hs_host[l];
    a.W = Ws_host[l];
    a.scale = a.L.scale[l];
    a.filter_lvl = l;
    if (int e = launch_bwd(a, (hipStream_t)stream)) return e;
  }
  return SD_OK;
}

