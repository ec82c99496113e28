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
#include "common.h"
#include "../../include/simpledet_ops.h"
#include <math.h>
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
  s.h_low = h_low + h_in; s.h_high = h_high + h_in;  // absolute rows / columns
  s.w_low = w_low + w_in; s.w_high = w_high + w_in;
  return s;
}

// grid: x = pixel tiles, y = (group, tap), z = image
__global__ __launch_bounds__(256) void deform_im2col_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ offset,
                                                            float* __restrict__ col, DcnGeom g) {
  const int P = g.Ho * g.Wo;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int K2 = g.kh * g.kw;
  const int grp = blockIdx.y / K2, tap = blockIdx.y % K2;
  const int i = tap / g.kw, j = tap % g.kw;
  const int n = blockIdx.z;
  const int cpg = g.C / g.dgroup;
  const int h_col = p / g.Wo, w_col = p % g.Wo;
  const int h_in = h_col * g.stride_h - g.pad_h, w_in = w_col * g.stride_w - g.pad_w;
  const float* off = offset + ((long)n * g.dgroup + grp) * 2 * K2 * P;
  const float offset_h = off[(long)(2 * tap) * P + p];
  const float offset_w = off[(long)(2 * tap + 1) * P + p];
  const Sample s = im2col_sample(g, h_in, w_in, i, j, offset_h, offset_w);
  const long plane = (long)g.H * g.W;
  const float* xc = x + ((long)n * g.C + (long)grp * cpg) * plane;
  float* out = col + (((long)n * g.C + (long)grp * cpg) * K2 + tap) * P + p;
  const int o1 = s.h_low * g.W + s.w_low, o2 = s.h_low * g.W + s.w_high;
  const int o3 = s.h_high * g.W + s.w_low, o4 = s.h_high * g.W + s.w_high;
#pragma unroll 4
  for (int c = 0; c < cpg; ++c) {
    float val = 0.f;
    if (s.ok) val = (s.w1 * xc[o1] + s.w2 * xc[o2] + s.w3 * xc[o3] + s.w4 * xc[o4]);
    *out = val;
    xc += plane;
    out += (long)K2 * P;
  }
}

__device__ __forceinline__ float get_gradient_weight(float argmax_h, float argmax_w, int h, int w,
                                                     int height, int width) {
  if (argmax_h < 0 || argmax_h > height || argmax_w < 0 || argmax_w > width) return 0;
  argmax_h = fmaxr(argmax_h, 0.f);
  argmax_w = fmaxr(argmax_w, 0.f);
  int argmax_h_low = (int)argmax_h, argmax_w_low = (int)argmax_w, argmax_h_high, argmax_w_high;
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

// Data gradient.  The reference scatters every col element with up to four global atomicAdds
// (620 M atomics for the (16,256,50,84) layer).  Here one workgroup owns a row band of ONE
// (image, channel) plane in LDS, accumulates the 9 taps x Ho*Wo col elements of that channel into
// it with LDS compare-and-swap adds and writes the band to HBM once: no global atomics, no
// zero-fill pass.  grid: x = channel, y = band, z = image.

__global__ __launch_bounds__(256) void deform_col2im_kernel(const float* __restrict__ col,
                                                            const float* __restrict__ offset,
                                                            float* __restrict__ dx, DcnGeom g,
                                                            int band_rows, int req_add) {
  extern __shared__ __attribute__((aligned(16))) float plane[];
  const int P = g.Ho * g.Wo, K2 = g.kh * g.kw;
  const int c = blockIdx.x, n = blockIdx.z;
  const int row0 = blockIdx.y * band_rows, row1 = iminr(row0 + band_rows, g.H);
  const int band_elems = (row1 - row0) * g.W;
  const int tid = threadIdx.x;
  for (int i = tid; i < band_elems; i += 256) plane[i] = 0.f;
  __syncthreads();
  const int cpg = g.C / g.dgroup, grp = c / cpg;
  const float* off = offset + ((long)n * g.dgroup + grp) * 2 * K2 * P;
  const float* cp = col + ((long)n * g.C + c) * K2 * P;
  // taps outer (wave-uniform), pixels inner with (h_out, w_out) advanced incrementally: no integer
  // division per col element
  const int step_h = 256 / g.Wo, step_w = 256 % g.Wo;
  for (int tap = 0; tap < K2; ++tap) {
   const int i = tap / g.kw, j = tap % g.kw;
   int h_out = tid / g.Wo, w_out = tid % g.Wo;
   for (int p = tid; p < P; p += 256, h_out += step_h, w_out += step_w) {
    if (w_out >= g.Wo) {
      w_out -= g.Wo;
      ++h_out;
    }
    const int idx = tap * P + p;
    const int h_in = h_out * g.stride_h - g.pad_h, w_in = w_out * g.stride_w - g.pad_w;
    const float offset_h = off[(2 * tap) * P + p];
    const float offset_w = off[(2 * tap + 1) * P + p];
    const float cur_inv_h_data = h_in + i * g.dil_h + offset_h;
    const float cur_inv_w_data = w_in + j * g.dil_w + offset_w;
    const int cur_h = (int)cur_inv_h_data;
    // quick reject: every touched row lies in [cur_h - 1, cur_h + 1]
    if (cur_h + 1 < row0 || cur_h - 1 >= row1) continue;
    const float cur_top_grad = cp[idx];
    // The reference walks the 5x5 neighbourhood of (cur_h, cur_w), keeps the pixels with
    // |inv_h - hh| < 1 and |inv_w - ww| < 1 and weighs them with get_gradient_weight().  Those
    // pixels are floor() and floor() + 1 of each coordinate (always inside the 5x5 window), and the
    // clamped corner rows / columns of get_gradient_weight() depend on the sample only: they are
    // computed once per col element, the per-pixel part is the factor selection of its if-chain.
    float ah = cur_inv_h_data, aw = cur_inv_w_data;
    if (ah < 0 || ah > g.H || aw < 0 || aw > g.W) continue;  // the function returns 0 for all pixels
    int hl = (int)ah, wl = (int)aw, hh_, wh_;
    if (hl >= g.H - 1) {
      hh_ = hl = g.H - 1;
      ah = (float)hl;
    } else {
      hh_ = hl + 1;
    }
    if (wl >= g.W - 1) {
      wh_ = wl = g.W - 1;
      aw = (float)wl;
    } else {
      wh_ = wl + 1;
    }
    const int fh = (int)floorf(cur_inv_h_data), fw = (int)floorf(cur_inv_w_data);
#pragma unroll
    for (int dy = 0; dy <= 1; dy++) {
      const int hh = fh + dy;
      if (!(hh >= 0 && hh < g.H && fabsf(cur_inv_h_data - hh) < 1)) continue;
      float fhv;
      if (hh == hl) fhv = (hh + 1 - ah);
      else if (hh == hh_) fhv = (ah + 1 - hh);
      else continue;
      if (!(hh >= row0 && hh < row1)) continue;
#pragma unroll
      for (int dxx = 0; dxx <= 1; dxx++) {
        const int ww = fw + dxx;
        if (!(ww >= 0 && ww < g.W && fabsf(cur_inv_w_data - ww) < 1)) continue;
        float fwv;
        if (ww == wl) fwv = (ww + 1 - aw);
        else if (ww == wh_) fwv = (aw + 1 - ww);
        else continue;
        const float w = fhv * fwv;
        if (w != 0.f) lds_add_cas(plane + (hh - row0) * g.W + ww, w * cur_top_grad);
      }
    }
   }
  }
  __syncthreads();
  float* d = dx + (((long)n * g.C + c) * g.H + row0) * g.W;
  for (int i = tid; i < band_elems; i += 256) d[i] = req_add ? d[i] + plane[i] : plane[i];
}

// The same gradient for CC channels of one deformable group at a time.  Where a col element lands
// and with which four weights depends on (tap, pixel, group) only, and working that out (the
// clamping chain of get_gradient_weight(), ~100 mostly divergent instructions) was the bulk of the
// kernel above, which repeats it for every channel.  Here a workgroup keeps the row band of CC
// channel planes in LDS, works the four (LDS index, weight) pairs out once per (tap, pixel) and
// applies them to the CC col values -- read as 16-byte loads, four pixels per lane.
//   grid: x = channel chunk, y = band, z = image; LDS = CC * band floats
//
// FX (round 4): the sums in 32-bit fixed point with plain integer LDS adds (fire and forget) instead of
// fp32 compare-and-swap loops -- 16 chained loops per sample were the kernel.  The unit needs a bound
// on what a pixel can collect: |dcol| <= cmax (the maximum the producing GEMM's epilogue recorded) times
// the largest sum of bilinear weights landing on one pixel, which depends on (image, group) only and
// is bounded per tap by deform_col2im_wsum_kernel (sum over the taps of each tap's largest pile-up).
// scale = the power of two that puts that bound below 2^29; one unit is then <= 2^-28 of the largest
// possible sum, and the result does not depend on the order of the adds.  A non-finite bound (inf /
// nan in dcol) keeps the compare-and-swap adds, which send inf / nan where the reference sends them.

constexpr unsigned kCmaxSlots = 32;   // words the producing GEMM spreads its max|C| over (GemmArgs::cmax)

// where a sample lands: LDS index of the (floor, floor) corner relative to the band and the factors of
// the two rows / two columns (0 for a corner that does not exist, lies outside the band or has weight 0)
__device__ __forceinline__ void col2im_geom(const DcnGeom& g, float inv_h, float inv_w, int row0, int row1,
                                            int& base, float (&fhv)[2], float (&fwv)[2]) {
  // same arithmetic as deform_col2im_kernel above, as (index, weight) pairs
  float ah = inv_h, aw = inv_w;
  const bool inside = !(ah < 0 || ah > g.H || aw < 0 || aw > g.W);
  int hl = (int)ah, wl = (int)aw, hh_, wh_;
  if (hl >= g.H - 1) {
    hh_ = hl = g.H - 1;
    ah = (float)hl;
  } else {
    hh_ = hl + 1;
  }
  if (wl >= g.W - 1) {
    wh_ = wl = g.W - 1;
    aw = (float)wl;
  } else {
    wh_ = wl + 1;
  }
  const int fh = (int)floorf(inv_h), fw = (int)floorf(inv_w);
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    const int hh = fh + d;
    const bool ok = inside && hh >= 0 && hh < g.H && fabsf(inv_h - hh) < 1 && hh >= row0 && hh < row1;
    fhv[d] = !ok ? 0.f : hh == hl ? (hh + 1 - ah) : hh == hh_ ? (ah + 1 - hh) : 0.f;
    const int ww = fw + d;
    const bool okw = ww >= 0 && ww < g.W && fabsf(inv_w - ww) < 1;
    fwv[d] = !okw ? 0.f : ww == wl ? (ww + 1 - aw) : ww == wh_ ? (aw + 1 - ww) : 0.f;
  }
  base = (fh - row0) * g.W + fw;
}

// per (tap, group, image): the largest sum of weights one pixel collects from this tap's samples, added
// into wsum[image * dgroup + group] (zeroed by the caller).  In integers -- every weight rounded UP to a
// multiple of 2^-wshift, integer LDS adds, integer maximum, integer sum over the taps -- so that the
// bound, and with it the fixed-point unit of the scatter, is the same in every run.  wshift is chosen
// by the host so that P weights of 1 cannot overflow 32 bits.  LDS = H * W words
__global__ __launch_bounds__(512) void deform_col2im_wsum_kernel(const float* __restrict__ offset,
                                                                 unsigned* __restrict__ wsum, DcnGeom g, int wshift) {
  extern __shared__ __attribute__((aligned(16))) unsigned wplane[];
  __shared__ unsigned s_max[8];
  const int P = g.Ho * g.Wo, K2 = g.kh * g.kw, HW = g.H * g.W;
  const int tap = blockIdx.x, grp = blockIdx.y, n = blockIdx.z, tid = threadIdx.x;
  for (int i = tid; i < HW; i += 512) wplane[i] = 0u;
  __syncthreads();
  const float* oh = offset + (((long)n * g.dgroup + grp) * 2 * K2 + 2 * tap) * P;
  const float* ow = oh + P;
  const int ti = tap / g.kw, tj = tap % g.kw;
  const float wscale = (float)(1u << wshift);
  for (int p = tid; p < P; p += 512) {
    const int h_out = p / g.Wo, w_out = p - h_out * g.Wo;
    const float inv_h = h_out * g.stride_h - g.pad_h + ti * g.dil_h + oh[p];
    const float inv_w = w_out * g.stride_w - g.pad_w + tj * g.dil_w + ow[p];
    int base;
    float fhv[2], fwv[2];
    col2im_geom(g, inv_h, inv_w, 0, g.H, base, fhv, fwv);
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const float w = fhv[d >> 1] * fwv[d & 1];
      if (w != 0.f)   // (w <= 1; NaN offsets give w == 0 through the comparisons of col2im_geom)
        __hip_atomic_fetch_add(wplane + base + (d >> 1) * g.W + (d & 1), (unsigned)ceilf(fminr(w, 1.f) * wscale),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  unsigned m = 0;
  for (int i = tid; i < HW; i += 512) m = m > wplane[i] ? m : wplane[i];
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const unsigned t = (unsigned)__shfl_xor((int)m, o);
    m = t > m ? t : m;
  }
  if ((tid & 63) == 0) s_max[tid >> 6] = m;
  __syncthreads();
  if (tid == 0) {
    unsigned t = 0;
    for (int k = 0; k < 8; ++k) t = s_max[k] > t ? s_max[k] : t;
    atomicAdd(wsum + (long)n * g.dgroup + grp, t);   // (<= K2 * P * 2^wshift < 2^32: the host's choice of wshift)
  }
}

template <int CC, int T, bool FX>
__global__ __launch_bounds__(T) void deform_col2im_chunk_kernel(const float* __restrict__ col,
                                                                const float* __restrict__ offset,
                                                                float* __restrict__ dx, DcnGeom g,
                                                                int band_rows, int req_add,
                                                                const unsigned* __restrict__ cmax,
                                                                const unsigned* __restrict__ wsum, int wshift) {
  extern __shared__ __attribute__((aligned(16))) float plane[];
  const int P = g.Ho * g.Wo, K2 = g.kh * g.kw;
  const int c0 = blockIdx.x * CC, n = blockIdx.z;
  const int row0 = blockIdx.y * band_rows, row1 = iminr(row0 + band_rows, g.H);
  const int band_elems = (row1 - row0) * g.W;
  const int tid = threadIdx.x;
  for (int i = tid; i < CC * band_elems; i += T) plane[i] = 0.f;   // (0.f and 0 are the same bits)
  __syncthreads();
  const int cpg = g.C / g.dgroup, grp = c0 / cpg;
  // fixed point: scale * (largest possible sum) < 2^29 (rounding of the individual adds and the slack of
  // the fp32 weight sums stay far inside the remaining two bits)
  bool fx = false;
  float scale = 1.f;
  if (FX) {
    // (the integer weight sum is exact in a float up to 2^24 units; beyond that it is rounded to nearest:
    // one more unit of margin)
    unsigned cbits = 0;
#pragma unroll
    for (unsigned k = 0; k < kCmaxSlots; ++k) cbits = cbits > cmax[k] ? cbits : cmax[k];
    const float bound = __uint_as_float(cbits) * ((float)(wsum[(long)n * g.dgroup + grp] + 1u) / (float)(1u << wshift)) * 1.000001f;
    const unsigned bb = __float_as_uint(bound);
    const int e = (int)((bb >> 23) & 255);
    if (bound == 0.f) {
      fx = true;   // nothing but zeros can arrive
    } else if (e != 255 && e != 0) {
      int es = 127 + 28 - (e - 127);   // scale = 2^(28 - floor(log2 bound))
      es = es > 254 ? 254 : es;
      if (es >= 1) {
        scale = __uint_as_float((unsigned)es << 23);
        fx = true;
      }
    }
  }
  const float* off = offset + ((long)n * g.dgroup + grp) * 2 * K2 * P;
  const float* cp = col + ((long)n * g.C + c0) * K2 * P;
  const long cstride = (long)K2 * P;  // col elements per channel
  for (int tap = 0; tap < K2; ++tap) {
    const int i = tap / g.kw, j = tap % g.kw;
    const float* oh = off + (long)(2 * tap) * P;
    const float* ow = oh + P;
    const float* ct = cp + (long)tap * P;
    for (int p4 = tid * 4; p4 < P; p4 += T * 4) {  // P % 4 == 0 (host)
      const float4 ofh = *reinterpret_cast<const float4*>(oh + p4);
      const float4 ofw = *reinterpret_cast<const float4*>(ow + p4);
      float4 cv[CC];
#pragma unroll
      for (int cc = 0; cc < CC; ++cc) cv[cc] = *reinterpret_cast<const float4*>(ct + cc * cstride + p4);
      int h_out = p4 / g.Wo, w_out = p4 - h_out * g.Wo;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float offset_h = e == 0 ? ofh.x : e == 1 ? ofh.y : e == 2 ? ofh.z : ofh.w;
        const float offset_w = e == 0 ? ofw.x : e == 1 ? ofw.y : e == 2 ? ofw.z : ofw.w;
        const int h_in = h_out * g.stride_h - g.pad_h, w_in = w_out * g.stride_w - g.pad_w;
        if (++w_out == g.Wo) {
          w_out = 0;
          ++h_out;
        }
        const float inv_h = h_in + i * g.dil_h + offset_h;
        const float inv_w = w_in + j * g.dil_w + offset_w;
        int base;
        float fhv[2], fwv[2];
        col2im_geom(g, inv_h, inv_w, row0, row1, base, fhv, fwv);
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const float w = fhv[d >> 1] * fwv[d & 1];
          if (w != 0.f) {
            const int idx = base + (d >> 1) * g.W + (d & 1);
            if (FX && fx) {
              // two channels per 64-bit add: (channel 2 k + 1) * 2^32 + (channel 2 k), the low field sign-extended;
              // both sums stay below 2^29 in magnitude, so the fields come apart again exactly (write-out)
              long long* q64 = reinterpret_cast<long long*>(plane) + idx;
#pragma unroll
              for (int pr = 0; pr < CC / 2; ++pr) {
                const float g0 = e == 0 ? cv[2 * pr].x : e == 1 ? cv[2 * pr].y : e == 2 ? cv[2 * pr].z : cv[2 * pr].w;
                const float g1 = e == 0 ? cv[2 * pr + 1].x : e == 1 ? cv[2 * pr + 1].y : e == 2 ? cv[2 * pr + 1].z : cv[2 * pr + 1].w;
                const long long lo = (long long)__float2int_rn((w * g0) * scale);
                const long long hi = (long long)__float2int_rn((w * g1) * scale);
                __hip_atomic_fetch_add(q64 + pr * band_elems, hi * 4294967296ll + lo, __ATOMIC_RELAXED,
                                       __HIP_MEMORY_SCOPE_WORKGROUP);
              }
            } else {
              float* q = plane + idx;
#pragma unroll
              for (int cc = 0; cc < CC; ++cc) {
                const float gv = e == 0 ? cv[cc].x : e == 1 ? cv[cc].y : e == 2 ? cv[cc].z : cv[cc].w;
                lds_add_cas(q + cc * band_elems, w * gv);
              }
            }
          }
        }
      }
    }
  }
  __syncthreads();
  const float unscale = 1.0f / scale;   // exact
  if (FX && fx) {
    static_assert(CC % 2 == 0, "fixed point: two channels per 64-bit word");
    for (int pr = 0; pr < CC / 2; ++pr) {
      float* d0 = dx + (((long)n * g.C + c0 + 2 * pr) * g.H + row0) * g.W;
      float* d1 = d0 + (long)g.H * g.W;
      const long long* pl = reinterpret_cast<const long long*>(plane) + pr * band_elems;
      for (int i = tid; i < band_elems; i += T) {
        const long long sum = pl[i];
        const int lo = (int)(unsigned)(sum & 0xffffffffll);
        const int hi = (int)((sum - (long long)lo) >> 32);
        const float v0 = (float)lo * unscale, v1 = (float)hi * unscale;
        d0[i] = req_add ? d0[i] + v0 : v0;
        d1[i] = req_add ? d1[i] + v1 : v1;
      }
    }
    return;
  }
  for (int cc = 0; cc < CC; ++cc) {
    float* d = dx + (((long)n * g.C + c0 + cc) * g.H + row0) * g.W;
    const float* pl = plane + cc * band_elems;
    for (int i = tid; i < band_elems; i += T) d[i] = req_add ? d[i] + pl[i] : pl[i];
  }
}

// grid: x = pixel tiles, y = offset channel (group, tap, dir), z = image
__global__ __launch_bounds__(256) void deform_col2im_coord_kernel(const float* __restrict__ col,
                                                                  const float* __restrict__ x,
                                                                  const float* __restrict__ offset,
                                                                  float* __restrict__ doff,
                                                                  DcnGeom g, int req_add) {
  const int P = g.Ho * g.Wo;
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= P) return;
  const int K2 = g.kh * g.kw;
  const int c_off = blockIdx.y;            // offset channel within the image
  const int grp = c_off / (2 * K2);
  const int offset_c = c_off - grp * 2 * K2;
  const int tap = offset_c / 2, bp_dir = offset_c % 2;
  const int i = tap / g.kw, j = tap % g.kw;
  const int n = blockIdx.z;
  const int cpg = g.C / g.dgroup;
  const int h_out = p / g.Wo, w_out = p % g.Wo;
  const int h_in = h_out * g.stride_h - g.pad_h, w_in = w_out * g.stride_w - g.pad_w;
  const float* off = offset + ((long)n * g.dgroup + grp) * 2 * K2 * P;
  const float offset_h = off[(long)(2 * tap) * P + p];
  const float offset_w = off[(long)(2 * tap + 1) * P + p];
  float inv_h = h_in + i * g.dil_h + offset_h;
  float inv_w = w_in + j * g.dil_w + offset_w;
  if (inv_h < 0 || inv_w < 0 || inv_h >= g.H || inv_w >= g.W) inv_h = inv_w = -1;
  // get_coordinate_weight: the neighbour indices / factors do not depend on the channel
  float val = 0.f;
  float argmax_h = inv_h, argmax_w = inv_w;
  const bool zero = argmax_h < 0 || argmax_h > g.H || argmax_w < 0 || argmax_w > g.W;
  if (!zero) {
    int hl = (int)argmax_h, wl = (int)argmax_w, hh, wh;
    if (hl >= g.H - 1) {
      hh = hl = g.H - 1;
      argmax_h = (float)hl;
    } else {
      hh = hl + 1;
    }
    if (wl >= g.W - 1) {
      wh = wl = g.W - 1;
      argmax_w = (float)wl;
    } else {
      wh = wl + 1;
    }
    float f1, f2, f3, f4;  // factors of im[hl,wl], im[hl,wh], im[hh,wl], im[hh,wh]
    if (bp_dir == 0) {
      f1 = -1 * (wl + 1 - argmax_w); f2 = -1 * (argmax_w - wl);
      f3 = (wl + 1 - argmax_w);      f4 = (argmax_w - wl);
    } else {
      f1 = -1 * (hl + 1 - argmax_h); f2 = (hl + 1 - argmax_h);
      f3 = -1 * (argmax_h - hl);     f4 = (argmax_h - hl);
    }
    const long plane = (long)g.H * g.W;
    const float* xc = x + ((long)n * g.C + (long)grp * cpg) * plane;
    const float* cp = col + (((long)n * g.C + (long)grp * cpg) * K2 + tap) * P + p;
    const int o1 = hl * g.W + wl, o2 = hl * g.W + wh, o3 = hh * g.W + wl, o4 = hh * g.W + wh;
    for (int c = 0; c < cpg; ++c) {
      float weight = 0;
      weight += f1 * xc[o1];
      weight += f2 * xc[o2];
      weight += f3 * xc[o3];
      weight += f4 * xc[o4];
      val += weight * *cp;
      xc += plane;
      cp += (long)K2 * P;
    }
  }
  float* out = doff + ((long)n * g.dgroup * 2 * K2 + c_off) * P + p;
  *out = req_add ? *out + val : val;
}


// ---- LDS-plane variants (the default when a channel plane fits in LDS and kh*kw <= 9) ----------
// The gathers are the expensive part of the per-lane kernels above (4 scattered 4-byte loads per
// col element through the vector L1).  Here a workgroup owns (image, deformable group, tile of T
// output pixels): every lane computes the sampling state of its pixel's kh*kw taps ONCE (packed
// corner index + the four bilinear weights, kept in registers for all channels), then the
// workgroup walks the group's channels: the channel plane is copied to LDS with coalesced 16-byte
// loads and each lane takes its 4 corners per tap from LDS.  x is read from HBM/L2 in full lines,
// the col stores stay contiguous along the pixel axis, and the bilinear expression is evaluated in
// the same order as before (bit-identical col).
constexpr int kDcnMaxTaps = 9;

__device__ __forceinline__ void dcn_stage_plane(float* xs, const float* __restrict__ xp, int plane,
                                                bool vec, int tid, int T) {
  if (vec) {
    const float4* s4 = reinterpret_cast<const float4*>(xp);
    float4* d4 = reinterpret_cast<float4*>(xs);
    for (int i = tid; i < plane / 4; i += T) d4[i] = s4[i];
  } else {
    for (int i = tid; i < plane; i += T) xs[i] = xp[i];
  }
}

// packed corner state: bits 0-27 index of (h_low, w_low), bit 28 w_high - w_low, bit 29
// h_high - h_low, bit 30 "inside the image"; 0 = outside (reads corner 0, contributes exactly 0)
constexpr int kDcnInside = 1 << 30;
__device__ __forceinline__ int dcn_pack(bool ok, int h_low, int w_low, int h_high, int w_high,
                                        int W) {
  if (!ok) return 0;
  return (h_low * W + w_low) | ((w_high - w_low) << 28) | ((h_high - h_low) << 29) | kDcnInside;
}

// The four corners of a packed sample from the staged plane: two adjacent-pair LDS reads (the
// compiler fuses each pair into one ds_read2_b32) and selects for the clamped border cases, where
// the reference reads the low corner again.  The unused neighbour may lie past the staged plane
// (the launch pads the LDS buffer by W + 1 floats); it is discarded by the select.
struct Corners {
  float x1, x2, x3, x4;
};
__device__ __forceinline__ Corners dcn_corners(const float* xs, int in, int W) {
  const int o1 = in & 0xfffffff;
  const float a = xs[o1], b = xs[o1 + 1];
  const float c = xs[o1 + W], d = xs[o1 + W + 1];
  const bool dw = (in >> 28) & 1, dh = (in >> 29) & 1;
  Corners r;
  r.x1 = a;
  r.x2 = dw ? b : a;
  r.x3 = dh ? c : a;
  r.x4 = dh ? (dw ? d : c) : r.x2;
  return r;
}

// The part of a channel plane this workgroup's samples can touch: the contiguous float range
// [min first corner, max last corner] over all inside taps of all lanes (exact, whatever the
// offsets are: wild offsets simply widen it to the whole plane).  Only that range is staged per
// channel -- with offsets of a few pixels a 256-pixel tile needs ~1/3 of a 50x84 plane.  The packed
// corner indices are rebased to the start of the range (a multiple of 4 floats when the 16-byte
// path is used).  rng: two ints of LDS.
__device__ __forceinline__ void dcn_window(int (&info)[kDcnMaxTaps], int W, int plane, int vec,
                                           int* rng, int tid, int& start, int& count) {
  if (vec & 2) {  // A/B: stage whole planes
    start = 0;
    count = plane;
    return;
  }
  if (tid == 0) {
    rng[0] = 0x7fffffff;
    rng[1] = -1;
  }
  __syncthreads();
  int lo = 0x7fffffff, hi = -1;
#pragma unroll
  for (int tap = 0; tap < kDcnMaxTaps; ++tap) {
    const int in = info[tap];
    if (in & kDcnInside) {
      const int o1 = in & 0xfffffff;
      lo = iminr(lo, o1);
      hi = imaxr(hi, o1 + (((in >> 29) & 1) ? W : 0) + ((in >> 28) & 1));
    }
  }
  if (hi >= 0) {
    atomicMin(&rng[0], lo);
    atomicMax(&rng[1], hi);
  }
  __syncthreads();
  lo = rng[0];
  hi = rng[1];
  if (hi < 0) {
    start = 0;
    count = 0;
    return;
  }
  start = (vec & 1) ? (lo & ~3) : lo;
  const int end = (vec & 1) ? iminr((hi + 4) & ~3, plane) : hi + 1;
  count = end - start;
#pragma unroll
  for (int tap = 0; tap < kDcnMaxTaps; ++tap)
    if (info[tap] & kDcnInside) info[tap] -= start;
}

// grid: x = pixel tiles, y = group * nsplit + channel split, z = image.  NT = kh*kw when known at
// compile time (9 for the reference's 3x3 layers), 0 = run-time tap count <= kDcnMaxTaps
template <int T, int NT>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(5, 8)))
void deform_im2col_lds_kernel(const float* __restrict__ x, const float* __restrict__ offset,
                              float* __restrict__ col, DcnGeom g, int nsplit, int vec, int nt) {
  extern __shared__ __attribute__((aligned(16))) float xs[];
  const int P = g.Ho * g.Wo, K2 = NT ? NT : g.kh * g.kw, plane = g.H * g.W;
  const int tid = threadIdx.x;
  const int p = blockIdx.x * T + tid;
  const bool live = p < P;
  const int grp = blockIdx.y / nsplit, cs = blockIdx.y % nsplit;
  const int n = blockIdx.z;
  const int cpg = g.C / g.dgroup;
  const int cchunk = (cpg + nsplit - 1) / nsplit;
  const int c0 = cs * cchunk, c1 = iminr(c0 + cchunk, cpg);
  int info[kDcnMaxTaps];
  float w1[kDcnMaxTaps], w2[kDcnMaxTaps], w3[kDcnMaxTaps], w4[kDcnMaxTaps];
  {
    const int h_col = live ? p / g.Wo : 0, w_col = live ? p % g.Wo : 0;
    const int h_in = h_col * g.stride_h - g.pad_h, w_in = w_col * g.stride_w - g.pad_w;
    const float* off = offset + ((long)n * g.dgroup + grp) * 2 * K2 * P + (live ? p : 0);
    float oh[kDcnMaxTaps], ow[kDcnMaxTaps];
#pragma unroll
    for (int tap = 0; tap < kDcnMaxTaps; ++tap) {  // all offset loads in flight together
      oh[tap] = ow[tap] = 0.f;
      if (tap < K2) {
        oh[tap] = off[(long)(2 * tap) * P];
        ow[tap] = off[(long)(2 * tap + 1) * P];
      }
    }
#pragma unroll
    for (int tap = 0; tap < kDcnMaxTaps; ++tap) {
      info[tap] = 0;
      w1[tap] = w2[tap] = w3[tap] = w4[tap] = 0.f;
      if (tap < K2) {
        const Sample s = im2col_sample(g, h_in, w_in, tap / g.kw, tap % g.kw, oh[tap], ow[tap]);
        info[tap] = dcn_pack(s.ok && live, s.h_low, s.w_low, s.h_high, s.w_high, g.W);
        w1[tap] = s.w1; w2[tap] = s.w2; w3[tap] = s.w3; w4[tap] = s.w4;
      }
      __builtin_amdgcn_sched_barrier(0);  // one tap's temporaries at a time (register pressure)
    }
  }
  int wstart, wcount;
  dcn_window(info, g.W, plane, vec, reinterpret_cast<int*>(xs + plane + g.W + 4), tid, wstart,
             wcount);
  for (int c = c0; c < c1; ++c) {
    const long ch = (long)n * g.C + (long)grp * cpg + c;
    __syncthreads();  // the previous channel's readers are done
    if (!(nt & 2)) dcn_stage_plane(xs, x + ch * plane + wstart, wcount, (vec & 1) != 0, tid, T);
    __syncthreads();
    float* out = col + ch * K2 * P;  // wave-uniform base + 32-bit lane offset
#pragma unroll
    for (int tap = 0; tap < kDcnMaxTaps; ++tap) {
      if (tap < K2) {
        asm volatile("" : "+v"(info[tap]));  // keep the unpacking inside the loop (registers)
        const int in = info[tap];
        const Corners q = dcn_corners(xs, in, g.W);
        float val = (w1[tap] * q.x1 + w2[tap] * q.x2 + w3[tap] * q.x3 + w4[tap] * q.x4);
        if (!(in & kDcnInside)) val = 0.f;
        if (live && !(nt & 4)) {
          if (nt & 1) __builtin_nontemporal_store(val, out + (tap * P + p));
          else out[tap * P + p] = val;
        }
        if ((nt & 4) && val == 12345.678f) out[0] = val;  // profiling only: keep the value alive
      }
    }
  }
}

// Offset gradient with the same ownership: both directions of a tap share the four corner values,
// the sum over the group's channels runs in registers in ascending channel order (as the per-lane
// kernel and the reference do).  grid: x = pixel tiles, y = group, z = image
template <int T, int NT>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(5, 8)))
void deform_col2im_coord_lds_kernel(const float* __restrict__ col, const float* __restrict__ x,
                                    const float* __restrict__ offset, float* __restrict__ doff,
                                    DcnGeom g, int req_add, int vec) {
  extern __shared__ __attribute__((aligned(16))) float xs[];
  const int P = g.Ho * g.Wo, K2 = NT ? NT : g.kh * g.kw, plane = g.H * g.W;
  const int tid = threadIdx.x;
  const int p = blockIdx.x * T + tid;
  const bool live = p < P;
  const int grp = blockIdx.y, n = blockIdx.z;
  const int cpg = g.C / g.dgroup;
  int info[kDcnMaxTaps];
  // (wl + 1 - aw) == 1 - (aw - wl) bit for bit (aw - wl is exact, both are one rounding of the
  // same real number), so only the two fractions are kept per tap
  float fb[kDcnMaxTaps], fd[kDcnMaxTaps];
  float val_h[kDcnMaxTaps], val_w[kDcnMaxTaps];
  {
    const int h_out = live ? p / g.Wo : 0, w_out = live ? p % g.Wo : 0;
    const int h_in = h_out * g.stride_h - g.pad_h, w_in = w_out * g.stride_w - g.pad_w;
    const float* off = offset + ((long)n * g.dgroup + grp) * 2 * K2 * P + (live ? p : 0);
    float oh[kDcnMaxTaps], ow[kDcnMaxTaps];
#pragma unroll
    for (int tap = 0; tap < kDcnMaxTaps; ++tap) {
      oh[tap] = ow[tap] = 0.f;
      if (tap < K2) {
        oh[tap] = off[(long)(2 * tap) * P];
        ow[tap] = off[(long)(2 * tap + 1) * P];
      }
    }
#pragma unroll
    for (int tap = 0; tap < kDcnMaxTaps; ++tap) {
      info[tap] = 0;
      fb[tap] = fd[tap] = 0.f;
      val_h[tap] = val_w[tap] = 0.f;
      if (tap < K2) {
        float inv_h = h_in + (tap / g.kw) * g.dil_h + oh[tap];
        float inv_w = w_in + (tap % g.kw) * g.dil_w + ow[tap];
        if (inv_h < 0 || inv_w < 0 || inv_h >= g.H || inv_w >= g.W) inv_h = inv_w = -1;
        float argmax_h = inv_h, argmax_w = inv_w;
        const bool zero = argmax_h < 0 || argmax_h > g.H || argmax_w < 0 || argmax_w > g.W;
        int hl = (int)argmax_h, wl = (int)argmax_w, hh, wh;
        if (hl >= g.H - 1) {
          hh = hl = g.H - 1;
          argmax_h = (float)hl;
        } else {
          hh = hl + 1;
        }
        if (wl >= g.W - 1) {
          wh = wl = g.W - 1;
          argmax_w = (float)wl;
        } else {
          wh = wl + 1;
        }
        info[tap] = dcn_pack(!zero && live, hl, wl, hh, wh, g.W);
        fb[tap] = (argmax_w - wl);  // direction h: -(1 - fb), -fb, +(1 - fb), +fb
        fd[tap] = (argmax_h - hl);  // direction w: -(1 - fd), +(1 - fd), -fd, +fd
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  int wstart, wcount;
  dcn_window(info, g.W, plane, vec, reinterpret_cast<int*>(xs + plane + g.W + 4), tid, wstart,
             wcount);
  for (int c = 0; c < cpg; ++c) {
    const long ch = (long)n * g.C + (long)grp * cpg + c;
    // the col values of this channel do not depend on the staged plane: issue their loads first
    // so that their latency overlaps the staging
    const float* cp = col + ch * K2 * P;  // wave-uniform base + 32-bit lane offset
    const int pl = live ? p : 0;
    float cv[kDcnMaxTaps];
#pragma unroll
    for (int tap = 0; tap < kDcnMaxTaps; ++tap) cv[tap] = tap < K2 ? cp[tap * P + pl] : 0.f;
    __syncthreads();
    dcn_stage_plane(xs, x + ch * plane + wstart, wcount, (vec & 1) != 0, tid, T);
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < kDcnMaxTaps; ++tap) {
      if (tap < K2) {
        // opaque to the optimiser: otherwise every product / select derived from the per-tap
        // state is hoisted out of the channel loop and the kernel spills
        asm volatile("" : "+v"(info[tap]), "+v"(fb[tap]), "+v"(fd[tap]));
        const int in = info[tap];
        const Corners q = dcn_corners(xs, in, g.W);
        const float x1 = q.x1, x2 = q.x2, x3 = q.x3, x4 = q.x4;
        const float fa = 1 - fb[tap], fc = 1 - fd[tap];
        float wh_ = 0;
        wh_ += (-1 * fa) * x1;
        wh_ += (-1 * fb[tap]) * x2;
        wh_ += fa * x3;
        wh_ += fb[tap] * x4;
        float ww_ = 0;
        ww_ += (-1 * fc) * x1;
        ww_ += fc * x2;
        ww_ += (-1 * fd[tap]) * x3;
        ww_ += fd[tap] * x4;
        if (in & kDcnInside) {
          val_h[tap] += wh_ * cv[tap];
          val_w[tap] += ww_ * cv[tap];
        }
      }
      if (tap % 3 == 2) __builtin_amdgcn_sched_barrier(0);  // three taps' corners in flight at most
    }
  }
  if (live) {
    float* out = doff + ((long)n * g.dgroup + grp) * 2 * K2 * P + p;
#pragma unroll
    for (int tap = 0; tap < kDcnMaxTaps; ++tap) {
      if (tap < K2) {
        float* oh = out + (long)(2 * tap) * P;
        float* ow = out + (long)(2 * tap + 1) * P;
        *oh = req_add ? *oh + val_h[tap] : val_h[tap];
        *ow = req_add ? *ow + val_w[tap] : val_w[tap];
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// fp32 MFMA GEMM:  C[b] (M x N) (+)= A[b] (M x K) . B[b] (K x N)
//   element A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]; one of each stride pair is 1.
// ------------------------------------------------------------------------------------------------
typedef float floatx16 __attribute__((ext_vector_type(16)));

struct GemmArgs {
  const float* A;
  const float* B;
  float* C;
  int M, N, K;
  long sam, sak, sbk, sbn;
  int ldc;
  long strideA, strideB, strideC;
  int mode;  // 0 store, 1 C += (read-modify-write), 2 atomic add
  int tiles_m, tiles_n;
  int fast;  // split kernel: operands aligned for vector loads (launch_gemm)
  long long* dbg;  // profiling build only: per-wave phase clocks
  long dbg_cap;
  int whole, whole_blocks, ksplit;  // split kernel: tiles run whole, their blocks (padded), k slices of the rest
  int ablate;  // profiling build only: bit 0 skip the A prefetch, bit 1 skip the B prefetch
  const unsigned* amax;  // f16 split: bit patterns of max|A|, max|B| (upper bounds), device memory
  unsigned* cmax;        // split kernel, optional: atomic max of the bit patterns of |C| as stored (an upper bound
                         // of max|C| when tiles are cut into k slices: slice maximum x slices)
};

constexpr int BM = 128;  // the N extent of a tile is 64 * J (J = 1, 2, 3), see launch_gemm

// Operand tile of ROWS rows x BK k-values, k-major in LDS.  Global element (r, k) sits at
// base[r*sr + k*sk] with one of the strides equal to 1.  NV = values per thread.
template <int ROWS, bool KCONTIG, int BK>
struct TileIO {
  // KCONTIG: unit = (row, 8 consecutive k): 2*ROWS units; else unit = (k, 4 consecutive rows): 4*ROWS
  static constexpr int SEGS = BK / 8;
  static constexpr int UNITS = KCONTIG ? SEGS * ROWS : (BK / 4) * ROWS;
  static constexpr int PER = KCONTIG ? 8 : 4;
  static constexpr int TRIPS = (UNITS + 255) / 256;
  static constexpr int NV = TRIPS * PER;
  static constexpr int LD = ROWS + 4;

  static __device__ __forceinline__ void load(const float* __restrict__ base, long sr, long sk,
                                              int r0, int k0, int R, int K, int tid,
                                              float (&v)[NV]) {
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) {
      const int u = tid + t * 256;
      if (UNITS % 256 != 0 && u >= UNITS) {
#pragma unroll
        for (int e = 0; e < PER; ++e) v[t * PER + e] = 0.f;
        continue;
      }
      if (KCONTIG) {
        const int r = r0 + u / SEGS, ks = k0 + (u % SEGS) * 8;
        const float* p = base + (long)r * sr + ks;
        if (r < R && ks + 7 < K && ((((uintptr_t)p) & 15) == 0)) {
          const float4 a = *reinterpret_cast<const float4*>(p);
          const float4 b = *reinterpret_cast<const float4*>(p + 4);
          v[t * 8 + 0] = a.x; v[t * 8 + 1] = a.y; v[t * 8 + 2] = a.z; v[t * 8 + 3] = a.w;
          v[t * 8 + 4] = b.x; v[t * 8 + 5] = b.y; v[t * 8 + 6] = b.z; v[t * 8 + 7] = b.w;
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[t * 8 + e] = (r < R && ks + e < K) ? p[e] : 0.f;
        }
      } else {
        constexpr int QR = ROWS / 4;  // 4-row groups per k row
        const int k = k0 + u / QR, rr = r0 + (u % QR) * 4;
        const float* p = base + (long)k * sk + rr;
        if (k < K && rr + 3 < R && ((((uintptr_t)p) & 15) == 0)) {
          const float4 a = *reinterpret_cast<const float4*>(p);
          v[t * 4 + 0] = a.x; v[t * 4 + 1] = a.y; v[t * 4 + 2] = a.z; v[t * 4 + 3] = a.w;
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[t * 4 + e] = (k < K && rr + e < R) ? p[e] : 0.f;
        }
      }
    }
  }

  static __device__ __forceinline__ void store(float* __restrict__ T, int tid, const float (&v)[NV]) {
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) {
      const int u = tid + t * 256;
      if (UNITS % 256 != 0 && u >= UNITS) continue;
      if (KCONTIG) {
        const int r = u / SEGS, ks = (u % SEGS) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) T[(ks + e) * LD + r] = v[t * 8 + e];
      } else {
        constexpr int QR = ROWS / 4;
        const int k = u / QR, rr = (u % QR) * 4;
        *reinterpret_cast<float4*>(&T[k * LD + rr]) =
            make_float4(v[t * 4], v[t * 4 + 1], v[t * 4 + 2], v[t * 4 + 3]);
      }
    }
  }
};

// 128 x (64*J) x 16 tiles, 4 waves as 2 x 2, each wave 64 x (32*J): 2 x J accumulators of 32x32
template <bool AK, bool BKC, int J, int BK>
__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(GemmArgs a) {
  using TA = TileIO<BM, AK, BK>;
  using TB = TileIO<64 * J, BKC, BK>;
  constexpr int BN = 64 * J;
  __shared__ __attribute__((aligned(16))) float As[BK * TA::LD];
  __shared__ __attribute__((aligned(16))) float Bs[BK * TB::LD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // tile order: consecutive blocks walk the M tiles of one N panel (the B/col panel stays hot)
  const int tm = blockIdx.x % a.tiles_m, tn = blockIdx.x / a.tiles_m;
  const int b = blockIdx.z;
  const float* A = a.A + (long)b * a.strideA;
  const float* B = a.B + (long)b * a.strideB;
  float* C = a.C + (long)b * a.strideC;
  const int m0 = tm * BM, n0 = tn * BN;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * (32 * J);  // the wave's 64 x 32J block

  floatx16 acc[2][J];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < J; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  float ra[TA::NV], rb[TB::NV];
  // A tile: rows = m, "row stride" sam, k stride sak.  B tile: rows = n, row stride sbn, k stride sbk
  TA::load(A, AK ? a.sam : 0, AK ? 1 : a.sak, m0, 0, a.M, a.K, tid, ra);
  TB::load(B, BKC ? a.sbn : 0, BKC ? 1 : a.sbk, n0, 0, a.N, a.K, tid, rb);
  for (int k0 = 0; k0 < a.K; k0 += BK) {
    __syncthreads();
    TA::store(As, tid, ra);
    TB::store(Bs, tid, rb);
    __syncthreads();
    if (k0 + BK < a.K) {
      TA::load(A, AK ? a.sam : 0, AK ? 1 : a.sak, m0, k0 + BK, a.M, a.K, tid, ra);
      TB::load(B, BKC ? a.sbn : 0, BKC ? 1 : a.sbk, n0, k0 + BK, a.N, a.K, tid, rb);
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      const int kr = kk + (lane >> 5);
      float av[2], bv[J];
#pragma unroll
      for (int i = 0; i < 2; ++i) av[i] = As[kr * TA::LD + wm + i * 32 + (lane & 31)];
#pragma unroll
      for (int j = 0; j < J; ++j) bv[j] = Bs[kr * TB::LD + wn + j * 32 + (lane & 31)];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < J; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
  }
  // D layout of the 32x32 tile: element e of lane l -> row (e/4)*8 + (l/32)*4 + e%4, col l%32
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const int col = n0 + wn + j * 32 + (lane & 31);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm + i * 32 + (e >> 2) * 8 + (lane >> 5) * 4 + (e & 3);
        if (row < a.M && col < a.N) {
          float* c = C + (long)row * a.ldc + col;
          const float v = acc[i][j][e];
          if (a.mode == 0) *c = v;
          else if (a.mode == 1) *c += v;
          else atomicAdd(c, v);
        }
      }
    }
}

// ------------------------------------------------------------------------------------------------
// The same product on the bf16 matrix cores: every fp32 operand is split into two bf16 parts
// (hi = RNE(x), lo = RNE(x - hi): 16 mantissa bits kept) while its tile is staged into LDS, and each
// fp32 product becomes  a_hi*b_hi + a_hi*b_lo + a_lo*b_hi  on v_mfma_f32_32x32x16_bf16 with fp32
// accumulation (the dropped a_lo*b_lo term is <= 2^-18 of the product).  Three bf16 MFMAs replace
// eight fp32 ones: 5.3x the fp32 MFMA peak.  Measured error of the DCN forward product (K = 2304):
// 4.5e-6 x max|C| against an fp64 product (plain fp32 accumulation: 5e-7) -- a twentieth of the
// 1e-4 parity bar.  Non-finite inputs give NaN (inf - inf in the split), as 0 x inf would.
//   tile 128 x 128 x 64, 4 waves as 2 x 2, each 64 x 64 = 2 x 2 accumulators of 32x32
//   LDS: four planes (A hi, A lo, B hi, B lo) of 128 rows x 64 bf16 (128 B per row, k contiguous);
//   the 16-byte granule gk (8 k values) of row r sits at position gk ^ ((r >> 1) & 7): fragment
//   reads (32 consecutive rows, one granule each: ds_read_b128) and both kinds of staging writes
//   (a row's 8 granules from 8 lanes; one granule of the even / odd rows from 64 lanes) touch all
//   32 banks evenly.
//   Tile order: the M tiles of one N panel run back to back on ONE XCD (block b -> XCD b % 8), so
//   the B panel (the col matrix, the only large operand) leaves HBM once.
// ------------------------------------------------------------------------------------------------
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

constexpr int SBM = 128, SBN = 128, SBK = 64;
constexpr int kSplitPlane = 128 * 128;  // bytes per LDS plane

// ---- split arithmetic ------------------------------------------------------------------------------
// kSplitBF16: hi = bf16(x), lo = bf16(x - hi): 16 mantissa bits kept, any fp32 magnitude.  Error of a
//   K = 2304 product sum ~4.5e-6 x max|C| (nine times the fp32 MFMA path's).
// kSplitF16 (default for the DCN layer): the operand is first scaled by a power of two s so that
//   max|x| * s lies in [2^13, 2^14) (s from a max|x| pre-pass over the operand -- or an upper bound of
//   it), then hi = f16(x * s), lo = f16(x * s - hi): 22 mantissa bits kept wherever |x| >= 2^-17 max|x|,
//   an absolute error below 2^-38 max|x| elsewhere (fp16 subnormals).  hi*hi products are exact in
//   the fp32 accumulator; the dropped lo*lo term is <= 2^-22 of a product.  The accumulator is scaled
//   back by 1/s_a and 1/s_b (exact) on the way out.  Measured: the error of the DCN products against
//   fp64 is that of the fp32 MFMA path (tests/test_deform_conv.py).
constexpr int kSplitBF16 = 1, kSplitF16 = 2;
typedef _Float16 halfx2 __attribute__((ext_vector_type(2)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));

// power-of-two scale of an operand from the bit pattern of (an upper bound of) its max|x|, and its
// inverse; zero, inf and nan maxima scale by 1
__device__ __forceinline__ void f16_split_scale(unsigned max_bits, float& s, float& inv) {
  const int e = (int)((max_bits >> 23) & 255);
  int es = 267 - e;  // biased exponent of s = 2^(13 - (e - 127))
  if ((max_bits & 0x7fffffffu) == 0u || e == 255) es = 127;
  es = es > 254 ? 254 : es;
  s = __uint_as_float((unsigned)es << 23);
  inv = 1.0f / s;  // exact: a power of two within the normal / subnormal range
}

// hi / lo parts of two floats, packed (element 0 in the low half).  PK = false keeps the two
// subtractions scalar: a packed v_pk_add_f32 wants its operands in adjacent registers, and for
// values that come out of two different loads the compiler then shuffles registers right behind
// the loads -- i.e. waits for the prefetch it was supposed to leave in flight.
template <bool PK, int MODE>
__device__ __forceinline__ void split2(float x0, float x1, float scale, unsigned& hi, unsigned& lo) {
  float f0, f1;
  if (MODE == kSplitF16) {
    x0 *= scale;
    x1 *= scale;
    const floatx2 v = {x0, x1};
    const halfx2 h = __builtin_convertvector(v, halfx2);
    hi = __builtin_bit_cast(unsigned, h);
    const floatx2 back = __builtin_convertvector(h, floatx2);
    f0 = back.x;
    f1 = back.y;
  } else {
    const floatx2 v = {x0, x1};
    const bf16x2 h = __builtin_convertvector(v, bf16x2);
    hi = __builtin_bit_cast(unsigned, h);
    f0 = __builtin_bit_cast(float, hi << 16);
    f1 = __builtin_bit_cast(float, hi & 0xffff0000u);
  }
  floatx2 r;
  if (PK) {
    r = floatx2{x0 - f0, x1 - f1};
  } else {
    float r0, r1;
    asm("v_sub_f32 %0, %1, %2" : "=v"(r0) : "v"(x0), "v"(f0));
    asm("v_sub_f32 %0, %1, %2" : "=v"(r1) : "v"(x1), "v"(f1));
    r = floatx2{r0, r1};
  }
  if (MODE == kSplitF16) lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, halfx2));
  else lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
}

// one 32x32x16 matrix-core step on packed 16-bit operands of either kind
template <int MODE>
__device__ __forceinline__ floatx16 mfma16(uint4 a, uint4 b, floatx16 c) {
  if (MODE == kSplitF16)
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(halfx8, a), __builtin_bit_cast(halfx8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// Operand tile of 128 rows x 64 k.  Global element (r, k) at base[r*sr + k*sk], one stride = 1.
template <bool KCONTIG>
struct SplitIO {
  // KCONTIG: unit = (row, granule of 8 k): 1024 units, 4 trips of 8 values (two 16-byte loads)
  // else   : unit = (4 consecutive rows, granule): 256 units, one trip of 32 values (eight 16-byte
  //          loads, one per k; a wave = 32 row quads x 2 granules: 512 contiguous bytes per k row)
  // Loads cost the wave ~100 cycles of issue each whatever their width, so all are 16 bytes wide.
  static constexpr int TRIPS = KCONTIG ? 4 : 1;
  static constexpr int NV = 32;

  // FAST (decided on the host for the whole launch): 16-byte (KCONTIG) / 8-byte aligned vector
  // loads of a full k step with no bounds tests -- rows past R are clamped (their products land in
  // rows / columns of C that are never stored).  The general version tests every element and is
  // used for the k tail and for unaligned operands.
  template <bool FAST>
  static __device__ __forceinline__ void load(const float* __restrict__ base, long sr, long sk,
                                              int r0, int k0, int R, int K, int tid,
                                              float (&v)[NV]) {
#pragma unroll
    for (int t = 0; t < TRIPS; ++t) {
      const int u = tid + t * 256;
      if (KCONTIG) {
        const int r = r0 + (u >> 3), ks = k0 + (u & 7) * 8;
        if (FAST) {
          const float* p = base + (long)min(r, R - 1) * sr + ks;
          const float4 a = *reinterpret_cast<const float4*>(p);
          const float4 b = *reinterpret_cast<const float4*>(p + 4);
          v[t * 8 + 0] = a.x; v[t * 8 + 1] = a.y; v[t * 8 + 2] = a.z; v[t * 8 + 3] = a.w;
          v[t * 8 + 4] = b.x; v[t * 8 + 5] = b.y; v[t * 8 + 6] = b.z; v[t * 8 + 7] = b.w;
        } else {
          const float* p = base + (long)r * sr + ks;
#pragma unroll
          for (int e = 0; e < 8; ++e) v[t * 8 + e] = (r < R && ks + e < K) ? p[e] : 0.f;
        }
      } else {
        const int rr = r0 + (u & 31) * 4, ks = k0 + (u >> 5) * 8;
        if (FAST) {  // R is a multiple of 4 here
          const float* p = base + (long)ks * sk + min(rr, R - 4);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            // kept as loaded (the registers of one load = rows rr .. rr + 3 of k = ks + e): a
            // shuffle here would wait for the data before the MFMAs the load is meant to hide under
            const float4 a = *reinterpret_cast<const float4*>(p + (long)e * sk);
            v[4 * e] = a.x; v[4 * e + 1] = a.y; v[4 * e + 2] = a.z; v[4 * e + 3] = a.w;
          }
        } else {
          const float* p = base + (long)ks * sk + rr;
#pragma unroll
          for (int e = 0; e < 8; ++e)
#pragma unroll
            for (int c = 0; c < 4; ++c) v[4 * e + c] = (ks + e < K && rr + c < R) ? p[(long)e * sk + c] : 0.f;
        }
      }
    }
  }

  // hi plane at T, lo plane at T + kSplitPlane (bytes)
  template <int MODE>
  static __device__ __forceinline__ void store(char* __restrict__ T, int tid, const float (&v)[NV], float scale) {
#pragma unroll
    for (int t = 0; t < NV / 8; ++t) {
      int r, gk;
      uint4 h, l;
      if (KCONTIG) {
        const int u = tid + t * 256;
        r = u >> 3;
        gk = u & 7;
        split2<true, MODE>(v[t * 8 + 0], v[t * 8 + 1], scale, h.x, l.x);
        split2<true, MODE>(v[t * 8 + 2], v[t * 8 + 3], scale, h.y, l.y);
        split2<true, MODE>(v[t * 8 + 4], v[t * 8 + 5], scale, h.z, l.z);
        split2<true, MODE>(v[t * 8 + 6], v[t * 8 + 7], scale, h.w, l.w);
      } else {  // row t of the quad: element k = e sits at v[4*e + t]
        r = (tid & 31) * 4 + t;
        gk = tid >> 5;
        split2<false, MODE>(v[t + 0], v[t + 4], scale, h.x, l.x);
        split2<false, MODE>(v[t + 8], v[t + 12], scale, h.y, l.y);
        split2<false, MODE>(v[t + 16], v[t + 20], scale, h.z, l.z);
        split2<false, MODE>(v[t + 24], v[t + 28], scale, h.w, l.w);
      }
      const int off = r * 128 + ((gk ^ ((r >> 1) & 7)) << 4);
      *reinterpret_cast<uint4*>(T + off) = h;
      *reinterpret_cast<uint4*>(T + kSplitPlane + off) = l;
    }
  }
};

template <bool AK, bool BKC, int MODE>
__global__ __launch_bounds__(256, 2) void gemm_f32_split_kernel(GemmArgs a) {
  using TA = SplitIO<AK>;
  using TB = SplitIO<BKC>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;                     // hi, lo
  char* Bs = smem + 2 * kSplitPlane;   // hi, lo
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Tile sequence q = ((image * tiles_n + tn) * tiles_m + tm).  Block b runs on XCD b % 8: an XCD
  // walks its N panels one after the other, all M tiles of each.  The first `whole` tiles of the
  // sequence are one block each; the rest (the tiles of a mostly empty last round of the resident
  // workgroups, launch_gemm) are cut into `ksplit` k slices that add into C atomically.
  int q, ks0 = 0, ks1 = 0x7fffffff;  // k-step range of this block
  bool piece = false;
  if ((int)blockIdx.x < a.whole_blocks) {
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    q = ((slot / a.tiles_m) * 8 + xcd) * a.tiles_m + slot % a.tiles_m;
    if (q >= a.whole) return;
  } else {
    const int pi = blockIdx.x - a.whole_blocks, sl = pi % a.ksplit;
    q = a.whole + pi / a.ksplit;
    const int nk = (a.K + SBK - 1) / SBK;
    ks0 = (int)((long)nk * sl / a.ksplit);
    ks1 = (int)((long)nk * (sl + 1) / a.ksplit);
    piece = true;
  }
  const int tm = q % a.tiles_m, tn = (q / a.tiles_m) % a.tiles_n;
  const int b = q / (a.tiles_m * a.tiles_n);
  const float* A = a.A + (long)b * a.strideA;
  const float* B = a.B + (long)b * a.strideB;
  float* C = a.C + (long)b * a.strideC;
  const int m0 = tm * SBM, n0 = tn * SBN;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

  floatx16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment addresses: row (lane & 31) of a 32-row block, granule 2*s + (lane >> 5)
  const int frow = lane & 31, fg = lane >> 5, fsw = (frow >> 1) & 7;
  const int a_off = (wm + frow) * 128, b_off = (wn + frow) * 128;

  auto mfma_step = [&]() {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const int go = ((2 * s + fg) ^ fsw) << 4;
      uint4 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = *reinterpret_cast<const uint4*>(As + a_off + i * 32 * 128 + go);
        al[i] = *reinterpret_cast<const uint4*>(As + kSplitPlane + a_off + i * 32 * 128 + go);
        bh[i] = *reinterpret_cast<const uint4*>(Bs + b_off + i * 32 * 128 + go);
        bl[i] = *reinterpret_cast<const uint4*>(Bs + kSplitPlane + b_off + i * 32 * 128 + go);
      }
      // small terms first, so that the large one meets the running sum last
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          acc[i][j] = mfma16<MODE>(al[i], bh[j], acc[i][j]);
          acc[i][j] = mfma16<MODE>(ah[i], bl[j], acc[i][j]);
          acc[i][j] = mfma16<MODE>(ah[i], bh[j], acc[i][j]);
        }
    }
  };
  const long sra = AK ? a.sam : 1, ska = AK ? 1 : a.sak, srb = BKC ? a.sbn : 1, skb = BKC ? 1 : a.sbk;
  float ra[32], rb[32];
  float sa = 1.f, sb = 1.f, inva = 1.f, invb = 1.f;
  if (MODE == kSplitF16) {
    f16_split_scale(a.amax[0], sa, inva);
    f16_split_scale(a.amax[1], sb, invb);
  }
#ifdef SD_PROFILING
  long long p_vm = 0, p_cvt = 0, p_mfma = 0, p_ld = 0;
  const long long p_begin = __builtin_readcyclecounter();
#endif
  // full k steps of aligned operands: register prefetch of the next tile under the MFMAs
  const int nfull = min(a.fast ? a.K / SBK : 0, ks1);
  if (nfull > ks0) {
    TA::template load<true>(A, sra, ska, m0, ks0 * SBK, a.M, a.K, tid, ra);
    TB::template load<true>(B, srb, skb, n0, ks0 * SBK, a.N, a.K, tid, rb);
    for (int s = ks0; s < nfull; ++s) {
#ifdef SD_PROFILING
      const long long c0 = __builtin_readcyclecounter();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const long long c1 = __builtin_readcyclecounter();
#endif
      __syncthreads();
      TA::template store<MODE>(As, tid, ra, sa);
      TB::template store<MODE>(Bs, tid, rb, sb);
      __syncthreads();
#ifdef SD_PROFILING
      const long long c2 = __builtin_readcyclecounter();
#endif
      {
        // the last step prefetches its own tile again (never used): no branch, so the loads and the
        // MFMAs below are one scheduling region and can be interleaved
        const int kn = min(s + 1, nfull - 1) * SBK;
#ifdef SD_PROFILING
        if (!(a.ablate & 1)) TA::template load<true>(A, sra, ska, m0, kn, a.M, a.K, tid, ra);
        if (!(a.ablate & 2)) TB::template load<true>(B, srb, skb, n0, kn, a.N, a.K, tid, rb);
#else
        TA::template load<true>(A, sra, ska, m0, kn, a.M, a.K, tid, ra);
        TB::template load<true>(B, srb, skb, n0, kn, a.N, a.K, tid, rb);
#endif
      }
#ifdef SD_PROFILING
      const long long c2a = c2;
#endif
      mfma_step();
      // one prefetch load per three MFMAs: a load costs the wave ~100 cycles of issue, which the
      // matrix pipe covers with the MFMAs already queued (issued as a block in front of the MFMAs,
      // the 16 loads stall the wave for ~1.7 k cycles with the pipe idle)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
      }
#ifdef SD_PROFILING
      __builtin_amdgcn_sched_barrier(0);
      const long long c3 = __builtin_readcyclecounter();
      p_vm += c1 - c0; p_cvt += c2 - c1; p_mfma += c3 - c2a; p_ld += c2a - c2;
#endif
    }
  }
#ifdef SD_PROFILING
  if (a.dbg && lane == 0) {
    const long id = (long)blockIdx.x * 4 + wave;
    if (id < a.dbg_cap) {
      long long* d = a.dbg + id * 8;
      d[0] = p_vm; d[1] = p_cvt; d[2] = p_mfma; d[3] = p_begin; d[4] = __builtin_readcyclecounter();
      d[5] = nfull - ks0; d[6] = p_ld;
    }
  }
#endif
  // the k tail, and everything when an operand is not aligned
  for (int k0 = max(nfull, ks0) * SBK; k0 < a.K && k0 < (long)ks1 * SBK; k0 += SBK) {
    TA::template load<false>(A, sra, ska, m0, k0, a.M, a.K, tid, ra);
    TB::template load<false>(B, srb, skb, n0, k0, a.N, a.K, tid, rb);
    __syncthreads();
    TA::template store<MODE>(As, tid, ra, sa);
    TB::template store<MODE>(Bs, tid, rb, sb);
    __syncthreads();
    mfma_step();
  }
  // D layout of the 32x32 tile: element e of lane l -> row (e/4)*8 + (l/32)*4 + e%4, col l%32
  unsigned vbits = 0;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + wn + j * 32 + (lane & 31);
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = m0 + wm + i * 32 + (e >> 2) * 8 + (lane >> 5) * 4 + (e & 3);
        if (row < a.M && col < a.N) {
          float* c = C + (long)row * a.ldc + col;
          float v = acc[i][j][e];
          if (MODE == kSplitF16) v = (v * inva) * invb;  // exact (powers of two)
          if (piece || a.mode == 2) atomicAdd(c, v);
          else if (a.mode == 0) *c = v;
          else *c += v;
          // (the maximum of the bit patterns of |v|: two integer operations per element, and a NaN -- whose
          // pattern lies above inf's -- survives as "not finite" where fmaxf would drop it)
          const unsigned ab = __float_as_uint(v) & 0x7fffffffu;
          vbits = vbits > ab ? vbits : ab;
        }
      }
    }
  if (a.cmax) {
    // one atomic per workgroup into one of kCmaxSlots words (tens of thousands of atomics on ONE word
    // serialise at the L2: +130 us on the col-gradient GEMM); readers take the maximum of the slots
    __shared__ float s_vmax[4];
    float vmax = __uint_as_float(vbits > 0x7f800000u ? 0x7f800000u : vbits);
    if (piece) vmax *= (float)a.ksplit;
    vmax = wave_max_f32(vmax);
    if (lane == 0) s_vmax[wave] = vmax;
    __syncthreads();
    if (tid == 0)
      atomicMax(a.cmax + ((unsigned)blockIdx.x % kCmaxSlots),
                __float_as_uint(fmaxr(fmaxr(s_vmax[0], s_vmax[1]), fmaxr(s_vmax[2], s_vmax[3]))));
  }
}

// zero the tiles that k slices add into (C = A.B with the last tiles cut along k)
__global__ __launch_bounds__(256) void gemm_zero_tiles_kernel(GemmArgs a) {
  const int q = a.whole + blockIdx.x;
  const int tm = q % a.tiles_m, tn = (q / a.tiles_m) % a.tiles_n, b = q / (a.tiles_m * a.tiles_n);
  float* C = a.C + (long)b * a.strideC;
  const int col = tn * SBN + (threadIdx.x & 127);
  if (col >= a.N) return;
  for (int r = threadIdx.x >> 7; r < SBM; r += 2) {
    const int row = tm * SBM + r;
    if (row < a.M) C[(long)row * a.ldc + col] = 0.f;
  }
}

template <bool AK, bool BKC, int MODE>
static int launch_gemm_split_m(const GemmArgs& g, dim3 grid, hipStream_t st) {
  // (every call: the attribute is per device, and a process may drive several)
  SD_HIP_CHECK(hipFuncSetAttribute((const void*)gemm_f32_split_kernel<AK, BKC, MODE>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, 4 * kSplitPlane));
  hipLaunchKernelGGL((gemm_f32_split_kernel<AK, BKC, MODE>), grid, dim3(256), 4 * kSplitPlane, st, g);
  return SD_OK;
}
template <bool AK, bool BKC>
static int launch_gemm_split(const GemmArgs& g, dim3 grid, hipStream_t st) {
  return g.amax ? launch_gemm_split_m<AK, BKC, kSplitF16>(g, grid, st)
                : launch_gemm_split_m<AK, BKC, kSplitBF16>(g, grid, st);
}

// Tile width by wave quantisation: the grid is only a few tiles per CU (4.1 for the DCN forward
// product with 128-wide tiles), so the last partial round costs up to a full tile time.  Pick the
// J in {1, 2, 3} that maximises  (tiles / CU) / ceil(tiles / CU)  x  (N / padded N)  x  the measured
// intrinsic rate of the variant (64- and 128-wide: ~102 TF on the DCN products, 6 and 3 waves per
// SIMD; 192-wide: ~96 TF, 2 waves per SIMD); ties go to the narrower tile (more waves resident).
template <bool AK, bool BKC>
static void launch_gemm_j(const GemmArgs& g, int J, dim3 grid, hipStream_t st) {
  if (J == 1) hipLaunchKernelGGL((gemm_f32_mfma_kernel<AK, BKC, 1, 16>), grid, dim3(256), 0, st, g);
  else if (J == 2) hipLaunchKernelGGL((gemm_f32_mfma_kernel<AK, BKC, 2, 16>), grid, dim3(256), 0, st, g);
  else hipLaunchKernelGGL((gemm_f32_mfma_kernel<AK, BKC, 3, 16>), grid, dim3(256), 0, st, g);
}

static int launch_gemm(GemmArgs& g, int batch, hipStream_t st) {
  if (g.M <= 0 || g.N <= 0 || batch <= 0) return SD_OK;
  // deform_gemm_split: 2 (default) scaled fp16 hi/lo split -- needs the operand maxima (g.amax: the
  // DCN entry points and sd_gemm_f32_ws provide them), without them the exact fp32 path runs;
  // 1 bf16 hi/lo split (no maxima needed, 9x the error); 0 fp32 MFMA
  int split = tuning("deform_gemm_split", 2);
  if (split == 2 && !g.amax) split = 0;
  if (split != 2) g.amax = nullptr;
  if (split >= 1) {
    const bool ak = g.sak == 1, bk = g.sbk == 1;
    SD_REQUIRE(ak || g.sam == 1, "GEMM: A needs a unit stride");
    SD_REQUIRE(bk || g.sbn == 1, "GEMM: B needs a unit stride");
    g.tiles_m = cdiv(g.M, SBM);
    g.tiles_n = cdiv(g.N, SBN);
    SD_REQUIRE((long)g.tiles_m * g.tiles_n * batch < (1L << 27), "GEMM: too many tiles");
    const int tiles = g.tiles_m * g.tiles_n * batch, nk = cdiv(g.K, SBK);
    // Two workgroups are resident per CU.  When the last round of them would be less than half
    // full, its tiles are cut into k slices (atomic adds into zeroed / existing C) so that the
    // round takes a slice's time instead of a tile's.  Sums of slices are order dependent in the
    // last bits; `deform_gemm_ksplit = 0` keeps every tile in one block.
    const int slots = 2 * kNumCU, rem = tiles % slots;
    g.whole = tiles;
    g.ksplit = 1;
    if (tuning("deform_gemm_ksplit", 1) == 1 && tiles > slots && rem > 0 && rem <= slots / 2 && nk >= 4) {
      int ks = slots / rem;
      if (ks > nk / 2) ks = nk / 2;
      if (ks >= 2) {
        g.whole = tiles - rem;
        g.ksplit = ks;
      }
    }
    const int groups = cdiv(g.whole, g.tiles_m);  // (image, N panel) groups of the whole tiles
    g.whole_blocks = cdiv(groups, 8) * 8 * g.tiles_m;
    const dim3 grid(g.whole_blocks + (tiles - g.whole) * g.ksplit, 1, 1);
    if (g.whole < tiles && g.mode == 0)
      hipLaunchKernelGGL(gemm_zero_tiles_kernel, dim3(tiles - g.whole), dim3(256), 0, st, g);
    auto aligned = [](const float* p, bool kc, long srow, long sk, long sbatch, int rows) {
      return kc ? (((uintptr_t)p & 15) == 0 && srow % 4 == 0 && sbatch % 4 == 0)
                : (((uintptr_t)p & 15) == 0 && sk % 4 == 0 && sbatch % 4 == 0 && rows % 4 == 0);
    };
#ifdef SD_PROFILING
    g.dbg = reinterpret_cast<long long*>(((uintptr_t)(unsigned)SD_PROF_TUNING("roi_align_dbg_hi", 0) << 32) |
                                         (uintptr_t)(unsigned)SD_PROF_TUNING("roi_align_dbg_lo", 0));
    g.dbg_cap = SD_PROF_TUNING("gemm_dbg_cap", 0);
    g.ablate = SD_PROF_TUNING("gemm_ablate", 0);
#endif
    g.fast = aligned(g.A, ak, g.sam, g.sak, g.strideA, g.M) &&
             aligned(g.B, bk, g.sbn, g.sbk, g.strideB, g.N);
    int e;
    if (ak && bk) e = launch_gemm_split<true, true>(g, grid, st);
    else if (ak) e = launch_gemm_split<true, false>(g, grid, st);
    else if (bk) e = launch_gemm_split<false, true>(g, grid, st);
    else e = launch_gemm_split<false, false>(g, grid, st);
    if (e) return e;
    SD_LAUNCH_CHECK();
    return SD_OK;
  }
  g.tiles_m = cdiv(g.M, BM);
  int J = 0;
  {
    double best = -1.0;
    for (int j = 1; j <= 3; ++j) {
      const int tn = cdiv(g.N, 64 * j);
      const double per_cu = (double)g.tiles_m * tn * batch / kNumCU;
      const double rounds = per_cu <= 1.0 ? 1.0 : (double)(long)(per_cu + 0.999999);
      double eff = (per_cu <= 1.0 ? per_cu : per_cu / rounds) * ((double)g.N / ((double)tn * 64 * j));
      if (j == 3) eff *= 0.94;
      if (eff > best) {
        best = eff;
        J = j;
      }
    }
  }
  g.tiles_n = cdiv(g.N, 64 * J);
  const dim3 grid(g.tiles_m * g.tiles_n, 1, batch);
  const bool ak = g.sak == 1, bk = g.sbk == 1;
  SD_REQUIRE(ak || g.sam == 1, "GEMM: A needs a unit stride");
  SD_REQUIRE(bk || g.sbn == 1, "GEMM: B needs a unit stride");
  if (ak && bk) launch_gemm_j<true, true>(g, J, grid, st);
  else if (ak) launch_gemm_j<true, false>(g, J, grid, st);
  else if (bk) launch_gemm_j<false, true>(g, J, grid, st);
  else launch_gemm_j<false, false>(g, J, grid, st);
  SD_LAUNCH_CHECK();
  return SD_OK;
}

// max|x| of a (batch, rows, cols) operand with row stride ld and batch stride bstride, as the bit
// pattern of the largest |x| (non-negative floats order like unsigned integers; a NaN wins, and the
// split then scales by 1): atomicMax into *out, which the caller zeroed.
struct AbsSeg {
  const float* p;
  long rows;
  int cols;
  long ld, bstride;
  int batch;
  unsigned* out;
  int blocks;   // workgroups of the launch that work on this operand
};

// workgroup `bid` of `nblk` on one operand
__device__ __forceinline__ void absmax_body(const AbsSeg& a, int bid, int nblk) {
  const float* __restrict__ p = a.p;
  const long rows = a.rows, ld = a.ld, bstride = a.bstride;
  const int cols = a.cols, batch = a.batch;
  const long per = rows * cols, n = per * batch;
  unsigned m = 0;
  const bool dense = ld == cols && (bstride == per || batch == 1) && (((uintptr_t)p & 15) == 0);
  if (dense) {
    const long n4 = n >> 2;
    const uint4* p4 = reinterpret_cast<const uint4*>(p);
    const long stride = (long)nblk * 256;
    long i = (long)bid * 256 + threadIdx.x;
    // four independent 16-byte loads in flight per lane (one dependent chain per lane runs at a
    // quarter of the HBM rate: 55 us for the 69 MB of x)
    for (; i + 3 * stride < n4; i += 4 * stride) {
      const uint4 v0 = p4[i], v1 = p4[i + stride], v2 = p4[i + 2 * stride], v3 = p4[i + 3 * stride];
      const unsigned a_ = max(max(v0.x & 0x7fffffffu, v0.y & 0x7fffffffu), max(v0.z & 0x7fffffffu, v0.w & 0x7fffffffu));
      const unsigned b_ = max(max(v1.x & 0x7fffffffu, v1.y & 0x7fffffffu), max(v1.z & 0x7fffffffu, v1.w & 0x7fffffffu));
      const unsigned c_ = max(max(v2.x & 0x7fffffffu, v2.y & 0x7fffffffu), max(v2.z & 0x7fffffffu, v2.w & 0x7fffffffu));
      const unsigned d_ = max(max(v3.x & 0x7fffffffu, v3.y & 0x7fffffffu), max(v3.z & 0x7fffffffu, v3.w & 0x7fffffffu));
      m = max(m, max(max(a_, b_), max(c_, d_)));
    }
    for (; i < n4; i += stride) {
      const uint4 v = p4[i];
      m = max(max(m, v.x & 0x7fffffffu), max(v.y & 0x7fffffffu, max(v.z & 0x7fffffffu, v.w & 0x7fffffffu)));
    }
    for (long i2 = (n4 << 2) + (long)bid * 256 + threadIdx.x; i2 < n; i2 += (long)nblk * 256)
      m = max(m, __float_as_uint(p[i2]) & 0x7fffffffu);
  } else {
    for (long i = (long)bid * 256 + threadIdx.x; i < n; i += (long)nblk * 256) {
      const long b = i / per, r = (i - b * per) / cols;
      const int c = (int)(i - b * per - r * cols);
      m = max(m, __float_as_uint(p[b * bstride + r * ld + c]) & 0x7fffffffu);
    }
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o));
  // one atomic per workgroup: thousands of same-address atomics cost more than the reads (measured:
  // 8192 of them 100 us, against 15 us for streaming the 69 MB)
  __shared__ unsigned wm[4];
  if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(wm[0], wm[1]), max(wm[2], wm[3]));
    if (m) atomicMax(a.out, m);
  }
}

// up to three operands in ONE launch (the layer's backward needs max|W|, max|dY| and max|x|: three
// launches of 5-15 us each plus their gaps otherwise)
__global__ __launch_bounds__(256) void absmax_kernel(AbsSeg s0, AbsSeg s1, AbsSeg s2) {
  const int b = blockIdx.x;
  if (b < s0.blocks) absmax_body(s0, b, s0.blocks);
  else if (b < s0.blocks + s1.blocks) absmax_body(s1, b - s0.blocks, s1.blocks);
  else absmax_body(s2, b - s0.blocks - s1.blocks, s2.blocks);
}

static AbsSeg absmax_seg(const float* p, long rows, int cols, long ld, long bstride, int batch, unsigned* out) {
  AbsSeg a{p, rows, cols, ld, bstride, batch, out, 0};
  const long n = rows * cols * batch;
  if (n <= 0 || !p) return a;
  long blocks = (n + 256 * 16 - 1) / (256 * 16);   // >= 16 floats per lane
  if (blocks > 4 * kNumCU) blocks = 4 * kNumCU;
  a.blocks = (int)blocks;
  return a;
}

static void launch_absmax(AbsSeg s0, AbsSeg s1 = AbsSeg{}, AbsSeg s2 = AbsSeg{}, hipStream_t st = nullptr) {
  const int blocks = s0.blocks + s1.blocks + s2.blocks;
  if (blocks <= 0) return;
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, s0, s1, s2);
}

// ------------------------------------------------------------------------------------------------
// Fused forward: y = W . col(x, offset) WITHOUT the col matrix (round 4).
//   The unfused forward writes col (620 MB for the (16,256,50,84) layer) and reads it back: ~12x the
//   bytes of x + offset + W + y.  Here a workgroup owns (image, tile of <= 96 output pixels) and ALL
//   F <= 256 filters, so every deformable sample is taken exactly once:
//     K order   k' = (half-slab of 8 channels, tap, channel): one k16 matrix-core step = two
//               consecutive (half-slab, tap) units, i.e. 9 steps per 16 channels.  A half-slab lies
//               inside one deformable group (C / dgroup % 16 == 0), so a unit's sampling state is ONE
//               packed corner index + four bilinear weights per pixel, computed once per (tile, group)
//               and kept in LDS (9 taps x 5 words per producer lane): the step loop is not unrolled
//               by tap, and the per-step cost is five ds_read_b32
//     x         the window of a half-slab's 8 channel planes that the tile's samples touch, in LDS;
//               two half-slab buffers form a ring: a buffer is refilled as soon as its last unit has
//               been sampled, four steps before its next use, by the FOURTH WAVE (which has no
//               sampling work) through its registers -- loads at the top of a step, LDS stores at its
//               end.  (global_load_lds fills would sit in front of every wave's A loads in the
//               in-order vmcnt queue with a count the compiler cannot know: a full drain per step.)
//     B tile    96 pixels x 16 k of one step: lanes 0..191 own (pixel, unit of the step), take the
//               corners from LDS, interpolate in fp32 with the im2col expression (the sampled values
//               are bit-equal to sd_deform_im2col's), scale + split into fp16 hi / lo and store two
//               16-byte granules; double-buffered, ONE workgroup barrier per step; the sampling of
//               step s + 1 is issued under the matrix-core ops of step s
//     A tile    the weights, pre-split once per call by dcn_prep_weight_kernel into the per-lane
//               fragment order of v_mfma_f32_32x32x16_f16; each wave loads its own 64 filter rows
//               straight from L2 into registers (no LDS, no VALU), two steps ahead
//     MFMA      4 waves x (64 filters x 96 pixels) = 2 x 3 accumulators of 32x32, three fp16 terms
//               per product (the scaled hi / lo split of the GEMM above)
//   Tiles per image are chosen so that the launch is a whole number of rounds of the 256 CUs
//   (one 256-thread workgroup per CU: the x windows take most of the LDS); tile t of image n runs on
//   XCD n % 8, so an image's planes are fetched into one L2.
//   A tile whose windows do not fit (wild offsets: 8 planes x window > 70 KB) flags itself and is
//   redone by the LDSX = false instance of the kernel, which takes its corners from global memory --
//   slow, but exact, and launched over the flagged tiles only.
// ------------------------------------------------------------------------------------------------
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// LDS slot of window position p: positions are permuted inside groups of four by the group's number,
// so that the loader's transposed stores (four consecutive groups at a time) land on 32 different banks
__device__ __forceinline__ int fslot(int p) { return p ^ ((p >> 2) & 3); }
constexpr int kFN = 96;            // pixel slots of a tile (3 x 32)
constexpr int kFHalf = 8;          // channels per half-slab
constexpr int kFThreads = 512;    // 4 matrix-core waves + 3 sampling waves + 1 loader wave
constexpr int kFBBytes = 2 * 2 * kFN * 16 * 2;   // B tile: 2 buffers x (hi, lo) x 96 x 16 halves = 12 KB
constexpr int kFStateBytes = 9 * 6 * kFN * 4;   // sampling state of 96 pixels x 9 taps x 6 words = 20.7 KB
constexpr int kFXFloats = 32512;   // 127 KB of x windows: two half-slab buffers
constexpr int kFStage = 16;        // 16-byte words per lane the loader wave moves per step (a quarter of a half-slab)
constexpr int kFSmemBytes = kFBBytes + kFStateBytes + kFXFloats * 4 + 64 + 64;

struct DcnFusedArgs {
  const float* x;
  const float* offset;
  const uint4* apre;     // pre-split weights, fragment order (dcn_prep_weight_kernel)
  const float* bias;     // (F) or null: added to y in the epilogue (`out += broadcast<1>(bias)`)
  float* y;
  DcnGeom g;
  int F, mtiles, nslab, tiles_per_image, tile_w;
  const unsigned* amax;  // {max|W|, max|x|}
  int x_aligned;         // x is 16-byte aligned (16-byte window loads); else every tile takes the global path
  int* flags;            // [tile] 1: left to the LDSX = false instance
  int ablate;            // profiling build only: 1 no sampling, 2 no matrix-core ops, 4 no window loads, 8 no B reads
  long long* dbg;        // profiling build only: per (workgroup, wave) {total, barrier wait, set-up} clocks
};

// weights (F, C, 9) -> apre[mt][slab16][j][wave][i][plane][lane] (16 bytes each): lane l of fragment
// (wave, i) holds filter row mt*256 + wave*64 + i*32 + (l & 31) and the 8 k values of unit
// u = 2 j + (l >> 5) of the slab: half-slab u / 9, tap u % 9, channels slab16*16 + (u / 9)*8 .. +7
__global__ __launch_bounds__(256) void dcn_prep_weight_kernel(const float* __restrict__ w, uint4* __restrict__ apre,
                                                              int F, int C, int mtiles, int nslab,
                                                              const unsigned* amax) {
  const long total = (long)mtiles * nslab * 9 * 4 * 2 * 64;   // (hi, lo) pairs
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= total) return;
  const int lane = (int)(e & 63);
  long r = e >> 6;
  const int i = (int)(r & 1); r >>= 1;
  const int wave = (int)(r & 3); r >>= 2;
  const int j = (int)(r % 9); r /= 9;
  const int slab = (int)(r % nslab);
  const int mt = (int)(r / nslab);
  const int f = mt * 256 + wave * 64 + i * 32 + (lane & 31);
  const int u = 2 * j + (lane >> 5), tap = u % 9;
  const int c0 = slab * 16 + (u / 9) * kFHalf;
  float s, inv;
  f16_split_scale(amax[0], s, inv);
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = (f < F && c0 + k < C) ? w[((long)f * C + c0 + k) * 9 + tap] : 0.f;
  uint4 h, l;
  split2<true, kSplitF16>(v[0], v[1], s, h.x, l.x);
  split2<true, kSplitF16>(v[2], v[3], s, h.y, l.y);
  split2<true, kSplitF16>(v[4], v[5], s, h.z, l.z);
  split2<true, kSplitF16>(v[6], v[7], s, h.w, l.w);
  const long base = ((((long)(mt * nslab + slab) * 9 + j) * 4 + wave) * 2 + i) * 2 * 64;
  apre[base + lane] = h;
  apre[base + 64 + lane] = l;
}

// dense copy of n4 16-byte words into LDS (destination = wave-uniform base + lane * 16)
__device__ __forceinline__ void dcn_fill16(const float* gsrc, int n4, float* dst, int wave, int lane) {
  const float4* s4 = reinterpret_cast<const float4*>(gsrc);
  float4* d4 = reinterpret_cast<float4*>(dst);
  for (int w4 = wave * 64; w4 < n4; w4 += (kFThreads / 64) * 64) {
    const int i = w4 + lane;
    if (i < n4) __builtin_amdgcn_global_load_lds(s4 + i, d4 + w4, 16, 0, 0);
  }
}

// workgroup barrier for LDS hand-overs only: waits for this wave's LDS traffic, NOT for its global
// loads (__syncthreads() carries a fence that drains vmcnt, i.e. every prefetch in flight)
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#ifdef SD_PROFILING
#define SD_FBAR()                                                   \
  do {                                                              \
    const long long t0_ = __builtin_readcyclecounter();             \
    lds_barrier();                                                  \
    p_wait += __builtin_readcyclecounter() - t0_;                   \
  } while (0)
#else
#define SD_FBAR() lds_barrier()
#endif

template <bool LDSX>
__global__ __launch_bounds__(kFThreads) void dcn_fwd_fused_kernel(DcnFusedArgs a) {
#ifdef SD_PROFILING
  long long p_wait = 0, p_setup = 0;
  const long long p_begin = __builtin_readcyclecounter();
#endif
  extern __shared__ __attribute__((aligned(16))) char fsm[];
  float* xs = reinterpret_cast<float*>(fsm);                              // two half-slab window buffers (at LDS
                                                                          // address 0: no base to add per read)
  char* Bs = fsm + kFXFloats * 4;
  float* sst = reinterpret_cast<float*>(fsm + kFXFloats * 4 + kFBBytes);  // sampling state [tap][word][pixel]
  int* rng = reinterpret_cast<int*>(fsm + kFXFloats * 4 + kFBBytes + kFStateBytes);
  const DcnGeom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int P = g.Ho * g.Wo, plane = g.H * g.W;
  // block -> (filter tile, image, pixel tile); the tiles of image n run on XCD n % 8
  const int per_mt = (int)gridDim.x / a.mtiles;
  const int mt = (int)blockIdx.x / per_mt, bb = (int)blockIdx.x % per_mt;
  const int xcd = bb & 7, slot = bb >> 3;
  const int n = (slot / a.tiles_per_image) * 8 + xcd, t = slot % a.tiles_per_image;
  if (n >= g.N) return;
  const int p0 = t * a.tile_w, p1 = iminr(p0 + a.tile_w, P);
  if (p0 >= P) return;
  int* flag = a.flags + ((long)mt * g.N + n) * a.tiles_per_image + t;
  if (!LDSX && *flag == 0) return;   // (the LDS instance has done this tile)
  const int cpg = g.C / g.dgroup;

  // ---- roles: waves 0..3 matrix cores (64 filter rows each), waves 4..6 sampling (192 lanes =
  // (pixel, unit of the step)), wave 7 the x windows.  One wave of the first kind and one of the
  // others share a SIMD: its matrix pipe and its vector ALU / LDS ports work side by side ----
  const int ptid = tid - 4 * 64;
  const bool producer = ptid >= 0 && ptid < 2 * kFN;
  const int pl = producer ? ptid % kFN : 0, half = producer ? ptid / kFN : 0;
  const int p = p0 + pl;
  const bool live = producer && p < p1;
  const int h_col = live ? p / g.Wo : 0, w_col = live ? p % g.Wo : 0;
  const int h_in = h_col * g.stride_h - g.pad_h, w_in = w_col * g.stride_w - g.pad_w;

  float sa, sb, inva, invb;
  f16_split_scale(a.amax[0], sa, inva);
  f16_split_scale(a.amax[1], sb, invb);

  const uint4* abase = a.apre + (long)mt * a.nslab * 9 * 1024 + (wave & 3) * 256 + lane;   // + step * 1024
  char* const bwr = Bs + half * (kFN * 16) + pl * 16;                          // this lane's B granule (hi)
  const char* const brd = Bs + (lane >> 5) * (kFN * 16) + (lane & 31) * 16;    // this lane's fragment rows
  const int nh = cpg / kFHalf, npair = nh / 2;   // half-slabs / 16-channel slabs of a group
  const int S = npair * 9;                       // steps of a group
  // Every workgroup walks the same k range, but starts somewhere else in it (group rot_g, then pair
  // rot_p of every group, wrapping around): 256 CUs reading the SAME 16 KB of pre-split weights in
  // the same step hammer a handful of L2 channels -- the A loads then take ~2000 clocks each
  // (measured: staggering the walk per workgroup lets the tiles of an image touch all of its channel
  // planes at once -- the image no longer fits its XCD's L2 and sigma = 2 offsets run 1.6x slower; off)
  const int rot_g = 0, rot_p = 0;
  auto grp_of = [&](int gi) { int v = gi + rot_g; return v >= g.dgroup ? v - g.dgroup : v; };
  auto pair_of = [&](int k) { int v = k + rot_p; return v >= npair ? v - npair : v; };   // k-th pair of the walk

  // ---- per group, every wave (same barriers in every role): the sampling state of the 9 taps into
  // LDS, the window the samples touch, the first two half-slabs.  false: the windows do not fit ----
  struct Grp { int wstart, wstride, n4; const float* xg; };
  auto group_begin = [&](int grp, Grp& G) -> bool {
    int wstart = 0, wcount = 0;
    {
      // the state of pixel pl is shared by its two lanes: lane (pl, 0) sets up taps 0..4, lane (pl, 1) taps 5..8
      constexpr int kT = 5;
      int info[kT];
      float w1[kT], w2[kT], w3[kT], w4[kT];
#pragma unroll
      for (int i = 0; i < kT; ++i) {
        info[i] = 0;
        w1[i] = w2[i] = w3[i] = w4[i] = 0.f;
      }
      const int tap0 = half * kT, ntap = half ? 9 - kT : kT;
      if (producer) {
        const float* off = a.offset + ((long)n * g.dgroup + grp) * 18 * P + (live ? p : 0);
        float oh[kT], ow[kT];
#pragma unroll
        for (int i = 0; i < kT; ++i) {
          const int tap = tap0 + (i < ntap ? i : 0);   // (lane (pl, 1)'s fifth slot: tap 5 again, not stored)
          oh[i] = off[(long)(2 * tap) * P];
          ow[i] = off[(long)(2 * tap + 1) * P];
        }
#pragma unroll
        for (int i = 0; i < kT; ++i) {
          const int tap = tap0 + (i < ntap ? i : 0);
          const int ti = (tap * 11) >> 5;   // tap / 3 for tap < 9
          const Sample s = im2col_sample(g, h_in, w_in, ti, tap - 3 * ti, oh[i], ow[i]);
          const bool in_ = s.ok && live && i < ntap;
          info[i] = dcn_pack(in_, s.h_low, s.w_low, s.h_high, s.w_high, g.W);
          // a sample outside the image (or a lane past the tile) has weights 0 and reads corner 0: the
          // sampling loop needs no "inside" select (0 x finite = 0; non-finite x gives NaN, as in the
          // split GEMM).  Likewise a column clamped at the border has lw == 0 exactly, so the weights of
          // the "right" corners are 0 and what is read there (the next row's first pixel, or the zeroed
          // slack behind the window) does not matter.
          w1[i] = in_ ? s.w1 : 0.f; w2[i] = in_ ? s.w2 : 0.f; w3[i] = in_ ? s.w3 : 0.f; w4[i] = in_ ? s.w4 : 0.f;
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      __syncthreads();   // (the previous group's last step is done with the B buffers, windows, state)
      {
        // the contiguous float range [first corner, last corner] over all inside taps of all lanes
        // (dcn_window's range, but reduced inside the wave first: a few hundred same-address LDS
        // atomics serialise), rebased to a multiple of four floats
        if (tid == 0) {
          rng[0] = 0x7fffffff;
          rng[1] = -1;
        }
        int lo = 0x7fffffff, hi = -1;
#pragma unroll
        for (int i = 0; i < kT; ++i) {
          const int in = info[i];
          if (in & kDcnInside) {
            const int o1 = in & 0xfffffff;
            lo = iminr(lo, o1);
            hi = imaxr(hi, o1 + (((in >> 29) & 1) ? g.W : 0) + ((in >> 28) & 1));
          }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
          lo = iminr(lo, __shfl_xor(lo, o));
          hi = imaxr(hi, __shfl_xor(hi, o));
        }
        __syncthreads();
        if (lane == 0 && hi >= 0) {
          atomicMin(&rng[0], lo);
          atomicMax(&rng[1], hi);
        }
        __syncthreads();
        lo = rng[0];
        hi = rng[1];
        if (hi >= 0) {
          wstart = lo & ~3;
          wcount = iminr((hi + 4) & ~3, plane) - wstart;
#pragma unroll
          for (int i = 0; i < kT; ++i)
            if (info[i] & kDcnInside) info[i] -= wstart;
        }
      }
      if (producer) {
        // state of (tap, pixel): three 8-byte words [tap][word][pixel].  Word 0 of the LDS instance: the
        // window slots of the four corners, 16 bits each (the window fits the LDS, or the tile is
        // flagged and this is never read); of the global-gather instance: the packed corner index
#pragma unroll
        for (int i = 0; i < kT; ++i)
          if (i < ntap) {
            int wa = info[i], wb = 0;
            if (LDSX) {
              const int o1 = info[i] & 0xfffffff, o2 = o1 + (((info[i] >> 29) & 1) ? g.W : 0);
              wa = fslot(o1) | (fslot(o1 + 1) << 16);
              wb = fslot(o2) | (fslot(o2 + 1) << 16);
            }
            float2* d = reinterpret_cast<float2*>(sst) + (tap0 + i) * 3 * kFN + pl;
            d[0] = make_float2(__int_as_float(wa), __int_as_float(wb));
            d[kFN] = make_float2(w1[i], w2[i]);
            d[2 * kFN] = make_float2(w3[i], w4[i]);
          }
      }
    }
    // positions behind a window in LDS: the second corner row of a sample may start W past the first
    // whatever the clamping, + 1 for the pair: a window is followed by W + 4 positions of slack
    G.wstart = wstart;
    G.wstride = (wcount + g.W + 4 + 3) & ~3;
    G.n4 = wcount >> 2;
    G.xg = a.x + ((long)n * g.C + (long)grp * cpg) * plane;   // channel 0 of the group
    if (LDSX) {
      // windows that do not fit two half-slab buffers, more words than the loader wave moves per
      // step, or a misaligned x: the tile is left to the global-gather instance
      // (and a non-finite x -- its maximum says so: the LDS instance reads corners it weighs with 0, an outside
      // sample's slot 0 or the neighbour behind a clamp, and 0 x inf would reach pixels the reference keeps
      // finite; the global-gather instance reads exactly what the reference reads)
      const bool fits = a.x_aligned && 2 * kFHalf * G.wstride <= kFXFloats && 2 * G.n4 <= 64 * kFStage &&
                        ((a.amax[1] >> 23) & 255u) != 255u;
      if (!fits) {
        if (tid == 0) *flag = 1;
        return false;
      }
      if (grp == 0 && tid == 0) *flag = 0;
      // the W + 4 .. W + 7 positions of slack behind the two windows are read (with weight 0) by samples
      // clamped at the border: keep them finite (whole groups of four positions: closed under fslot())
      {
        const int sl_ = kFHalf * (G.wstride - (G.n4 << 2));
        for (int i = tid; i < 2 * sl_; i += kFThreads)
          xs[(i / sl_) * kFHalf * G.wstride + kFHalf * (G.n4 << 2) + i % sl_] = 0.f;
      }
      // the first two half-slabs, by everybody, once per group
      {
        // lane = (channel quad l & 1, position group l >> 1), as in the loader wave below; the 16 waves'
        // worth of (half-slab, group) items are dealt round-robin
        const int cq = lane & 1;
        for (int it = wave; it < 2 * ((G.n4 + 31) >> 5); it += kFThreads / 64) {
          const int hs = it & 1, i = (it >> 1) * 32 + (lane >> 1);
          if (i < G.n4) {
            const float* src = G.xg + (long)(pair_of(0) * 2 * kFHalf + hs * kFHalf + cq * 4) * plane + wstart + 4 * i;
            f32x4 v[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) v[c] = *reinterpret_cast<const f32x4*>(src + (long)c * plane);
            float* d = xs + hs * kFHalf * G.wstride + (4 * i) * kFHalf + cq * 4;
            const int x_ = i & 3;
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
              f32x4 o; o[0] = v[0][jj]; o[1] = v[1][jj]; o[2] = v[2][jj]; o[3] = v[3][jj];
              *reinterpret_cast<f32x4*>(d + (jj ^ x_) * kFHalf) = o;
            }
          }
        }
      }
    }
    // the windows and the state landed (every wave waits for its own fill loads)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    return true;
  };

  if (wave < 4) {
    // ================= matrix-core waves: acc += A(step) . B(step) ====================================
    floatx16 acc[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    uint4 acur[4], anxt[4];
    auto load_a = [&](uint4 (&dst)[4], int grp, int k, int j) {   // A of step j of the k-th pair of the walk
      if (j >= 9) { j -= 9; ++k; }
      const int kk = k < npair ? k : npair - 1;   // (past the group's end: prefetched in vain, no branch)
      const uint4* q = abase + (long)((grp * npair + pair_of(kk)) * 9 + j) * 1024;
#pragma unroll
      for (int i = 0; i < 4; ++i) dst[i] = q[i * 64];
    };
    for (int gi = 0; gi < g.dgroup; ++gi) {
      const int grp = grp_of(gi);
      Grp G;
#ifdef SD_PROFILING
      const long long ts_ = __builtin_readcyclecounter();
#endif
      if (!group_begin(grp, G)) return;
#ifdef SD_PROFILING
      p_setup += __builtin_readcyclecounter() - ts_;
#endif
      load_a(anxt, grp, 0, 0);
      SD_FBAR();   // (the producers' B(0))
      // One step behind the B tiles: in interval s the fragments of B(s) are read (their LDS latency,
      // behind the sampling waves' reads in the same queue, is hidden) while the matrix cores work on
      // step s - 1 from registers.  B(s) is in registers by the interval's barrier, so its LDS buffer is
      // free for B(s + 2) exactly as before.
      uint4 bh[3], bl[3], nh_[3], nl_[3];
      auto read_b = [&](int st, uint4 (&h)[3], uint4 (&l)[3]) {
        const char* bb_ = brd + (st & 1) * (kFBBytes / 2);
#ifdef SD_PROFILING
        if (a.ablate & 16) {
#pragma unroll
          for (int q = 0; q < 3; ++q) h[q] = l[q] = acur[q];
          return;
        }
#endif
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          h[q] = *reinterpret_cast<const uint4*>(bb_ + q * 512);
          l[q] = *reinterpret_cast<const uint4*>(bb_ + kFN * 32 + q * 512);
        }
      };
      auto mma = [&]() {
#ifdef SD_PROFILING
        if (a.ablate & 2) return;
#endif
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int q = 0; q < 3; ++q) {
            acc[i][q] = mfma16<kSplitF16>(acur[2 * i + 1], bh[q], acc[i][q]);   // a_lo * b_hi
            acc[i][q] = mfma16<kSplitF16>(acur[2 * i], bl[q], acc[i][q]);       // a_hi * b_lo
            acc[i][q] = mfma16<kSplitF16>(acur[2 * i], bh[q], acc[i][q]);       // a_hi * b_hi
          }
      };
      auto next_a = [&](int k, int j) {   // acur = A(step), then the load of A(step + 1) goes out
#pragma unroll
        for (int i = 0; i < 4; ++i) acur[i] = anxt[i];
#ifdef SD_PROFILING
        if (!(a.ablate & 8))
#endif
        load_a(anxt, grp, k, j + 1);
      };
      int k = 0, j = 0;
      read_b(0, nh_, nl_);
      next_a(k, j);
      if (++j == 9) { j = 0; ++k; }
      if (S > 1) SD_FBAR();
      for (int s = 1; s < S; ++s) {
#pragma unroll
        for (int q = 0; q < 3; ++q) { bh[q] = nh_[q]; bl[q] = nl_[q]; }
        read_b(s, nh_, nl_);
        mma();             // step s - 1
        next_a(k, j);      // A(s)
        if (s + 1 < S) SD_FBAR();
        if (++j == 9) { j = 0; ++k; }
      }
#pragma unroll
      for (int q = 0; q < 3; ++q) { bh[q] = nh_[q]; bl[q] = nl_[q]; }
      mma();               // step S - 1
    }
    // ---- y[n, f, p] = acc / (s_w s_x): D layout of a 32x32 tile: element e of lane l -> row
    // (e / 4) * 8 + (l / 32) * 4 + e % 4, column l % 32 ----
    float* yn = a.y + (long)n * a.F * P;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int pp = p0 + j * 32 + (lane & 31);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int f = mt * 256 + wave * 64 + i * 32 + (e >> 2) * 8 + (lane >> 5) * 4 + (e & 3);
          if (f < a.F && pp < p1) {
            const float v = (acc[i][j][e] * inva) * invb;
            yn[(long)f * P + pp] = a.bias ? v + a.bias[f] : v;
          }
        }
      }
  } else if (wave < 7) {
    // ================= sampling waves: B(s + 1) while the matrix cores work on B(s) ===================
    for (int gi = 0; gi < g.dgroup; ++gi) {
      const int grp = grp_of(gi);
      Grp G;
#ifdef SD_PROFILING
      const long long ts_ = __builtin_readcyclecounter();
#endif
      if (!group_begin(grp, G)) return;
#ifdef SD_PROFILING
      p_setup += __builtin_readcyclecounter() - ts_;
#endif
      const int wstride = G.wstride, wstart = G.wstart;
      const float* xg = G.xg;
      // unit u = 2 j + half of the pair's step j: half-slab (= window buffer) u / 9, tap u % 9
      struct St { int wa, wb, hl; float a1, a2, a3, a4; };     // one unit's sampling state (hl: the window buffer)
      struct Rd { f32x4 va[2], vb[2], vc[2], vd[2]; St st; };  // its four corners x eight channels
      auto read_state = [&](int j, St& st) {
        const int u = 2 * j + half, hl = u >= 9 ? 1 : 0, tap = u - 9 * hl;
        const float2* sp = reinterpret_cast<const float2*>(sst) + tap * 3 * kFN + pl;
        const float2 q0 = sp[0], q1 = sp[kFN], q2 = sp[2 * kFN];
        st.wa = __float_as_int(q0.x); st.wb = __float_as_int(q0.y); st.hl = hl;
        st.a1 = q1.x; st.a2 = q1.y; st.a3 = q2.x; st.a4 = q2.y;
      };
      // the eight 16-byte reads of one unit (a corner's eight channels lie side by side) go out ...
      auto issue = [&](int pair, const St& st, Rd& r) {
        r.st = st;
        if (LDSX) {
          const char* xb = reinterpret_cast<const char*>(xs) + st.hl * (kFHalf * 4 * wstride);
          const f32x4* c1 = reinterpret_cast<const f32x4*>(xb + (st.wa & 0xffff) * (kFHalf * 4));
          const f32x4* c2 = reinterpret_cast<const f32x4*>(xb + ((unsigned)st.wa >> 16) * (kFHalf * 4));
          const f32x4* c3 = reinterpret_cast<const f32x4*>(xb + (st.wb & 0xffff) * (kFHalf * 4));
          const f32x4* c4 = reinterpret_cast<const f32x4*>(xb + ((unsigned)st.wb >> 16) * (kFHalf * 4));
          r.va[0] = c1[0]; r.va[1] = c1[1]; r.vb[0] = c2[0]; r.vb[1] = c2[1];
          r.vc[0] = c3[0]; r.vc[1] = c3[1]; r.vd[0] = c4[0]; r.vd[1] = c4[1];
        } else {
          // corners straight from global memory, every address inside the plane (window-relative
          // index made absolute, no "+ 1" past a clamp)
          const int tin = st.wa;
          const int o1 = tin & 0xfffffff;
          const int o2 = o1 + (((tin >> 29) & 1) ? g.W : 0);   // second corner row (the first again when clamped)
          const bool inside = (tin & kDcnInside) != 0;
          const float* xc = xg + (long)((2 * pair + st.hl) * kFHalf) * plane;
          const int g1 = inside ? o1 + wstart : 0, g2 = inside ? o2 + wstart : 0, d1 = (tin >> 28) & 1;
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            // (an outside sample contributes exactly 0 whatever x holds at index 0)
            const float ta = xc[(long)c * plane + g1], tb = xc[(long)c * plane + g1 + d1];
            const float tc = xc[(long)c * plane + g2], td = xc[(long)c * plane + g2 + d1];
            r.va[c >> 2][c & 3] = inside ? ta : 0.f; r.vb[c >> 2][c & 3] = inside ? tb : 0.f;
            r.vc[c >> 2][c & 3] = inside ? tc : 0.f; r.vd[c >> 2][c & 3] = inside ? td : 0.f;
          }
        }
      };
      // ... and are consumed one step later: interpolate (fused multiply-adds: within an ulp of
      // sd_deform_im2col's value), scale + split into fp16 hi / lo, store the two granules of B(step)
      auto finish = [&](const Rd& r, int step) {
        uint4 h4, l4;
        unsigned* hp = reinterpret_cast<unsigned*>(&h4);
        unsigned* lp = reinterpret_cast<unsigned*>(&l4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[2];
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int c = 2 * q + k;
            v[k] = __builtin_fmaf(r.st.a4, r.vd[c >> 2][c & 3], __builtin_fmaf(r.st.a3, r.vc[c >> 2][c & 3],
                                  __builtin_fmaf(r.st.a2, r.vb[c >> 2][c & 3], r.st.a1 * r.va[c >> 2][c & 3])));
          }
          split2<true, kSplitF16>(v[0], v[1], sb, hp[q], lp[q]);
        }
        char* bd = bwr + (step & 1) * (kFBBytes / 2);
        *reinterpret_cast<uint4*>(bd) = h4;
        *reinterpret_cast<uint4*>(bd + kFN * 32) = l4;
      };
      // Pipeline: in iteration s the reads of step s + 2 are issued first, then step s + 1 (read in
      // iteration s - 1) is finished under their latency; the state of step s + 3 is fetched behind them.
      //   (pj, pk): (step in pair, pair) of the step whose STATE is fetched next
      Rd ra, rb;
      St st;
      int pj = 0, pk = 0;
      auto next_state = [&]() {   // the state of the next step of the walk (past the end: the last step again)
        read_state(pj, st);
        if (pk * 9 + pj + 1 < S) { if (++pj == 9) { pj = 0; ++pk; } }
      };
      auto pair_now = [&](int step) { const int k_ = step / 9; return pair_of(k_ < npair ? k_ : npair - 1); };
      next_state();                        // state(0)
      issue(pair_now(0), st, ra);          // reads(0)
      next_state();                        // state(1)
      finish(ra, 0);                       // B(0)
      issue(pair_now(1), st, rb);          // reads(1)   (S >= 9: step 1 exists)
      next_state();                        // state(2)
      SD_FBAR();
      for (int s = 0;;) {   // S - 1 iterations (= barriers), two per trip: the read buffers alternate
        if (s + 1 >= S) break;
        issue(pair_now(s + 2), st, ra);   // reads(s + 2)   (past the end: the last step again, unused)
        next_state();
        __builtin_amdgcn_sched_barrier(0);   // (all reads out before the arithmetic on the other buffer starts)
#ifdef SD_PROFILING
        if (!(a.ablate & 1))
#endif
        finish(rb, s + 1);                // B(s + 1)
        SD_FBAR();
        ++s;
        if (s + 1 >= S) break;
        issue(pair_now(s + 2), st, rb);
        next_state();
        __builtin_amdgcn_sched_barrier(0);
#ifdef SD_PROFILING
        if (!(a.ablate & 1))
#endif
        finish(ra, s + 1);
        SD_FBAR();
        ++s;
      }
    }
  } else {
    // ================= loader wave: the ring of half-slab windows =====================================
    // One piece = a quarter of a half-slab's positions (all eight channels) per step, loaded into
    // registers in step s and stored to LDS at the top of step s + 1 (a whole step hides the load
    // latency; the step barrier never waits for memory).  Lane = (channel quad l & 1, position group
    // l >> 1): four wave-wide loads take 512 contiguous bytes of each of the quad's four channel
    // planes, and a lane's 4 x 4 block goes to LDS transposed, as four 16-byte stores of one position's
    // four channels into [slot][channel] (fslot() spreads the 8 lanes of a store phase over the 32
    // banks).  The schedule follows from when the sampling waves read a buffer last (see the step
    // loop below).
    // (measured alternatives on the channel-planar layout: global memory straight to LDS -- a half-slab
    // at once, or three channel windows per step -- is slower, 0.48 against 0.45 ms: a single wave
    // issues those at ~100 clocks each; as 4-byte pieces, 0.70 ms)
    for (int gi = 0; gi < g.dgroup; ++gi) {
      const int grp = grp_of(gi);
      Grp G;
#ifdef SD_PROFILING
      const long long ts_ = __builtin_readcyclecounter();
#endif
      if (!group_begin(grp, G)) return;
#ifdef SD_PROFILING
      p_setup += __builtin_readcyclecounter() - ts_;
#endif
      const int wstride = G.wstride, n4 = G.n4, Q = (n4 + 3) >> 2;   // Q: four-position groups of a piece
      const int cq = lane & 1, lg = lane >> 1;
      SD_FBAR();   // (the producers' B(0))
      // Two pieces in flight: the piece loaded in iteration s is stored at the top of iteration s + 2
      // from the register set of s's parity (two named sets and a loop unrolled by two: a
      // run-time-indexed set would live in scratch memory).
      //   (the sampling waves issue the reads of step X in iteration X - 2 and have them back by that
      //   iteration's barrier: buffer 0, last read for step 4, may be overwritten from iteration 3 on
      //   and must be complete by the end of iteration 6; buffer 1, last read for step 8, from
      //   iteration 7 on, complete by the end of the next pair's iteration 1.  Stores at the tops of
      //   iterations 3..6 and 7, 8, 0', 1': loads in iterations 1..4 and 5..8.)
      struct Pc { u32x4 stg[kFStage]; float* pend; int pend_i; };   // stg: [unit of 32 groups][channel of the quad]
      Pc pa, pb;
      pa.pend = pb.pend = nullptr;   // where the piece goes: slot 4 * (first group), this lane's channel quad
      pa.pend_i = pb.pend_i = 0;     // its first group + lg
      auto lstep = [&](Pc& pc, int pair, int j) {
        if (pc.pend) {
          const int x_ = pc.pend_i & 3;
          float* d = pc.pend + lg * (4 * kFHalf);
#pragma unroll
          for (int u = 0; u < kFStage / 4; ++u) {
            if (32 * u >= Q) break;   // (wave-uniform: a piece of a small window has fewer units)
            if (32 * u + lg < Q && pc.pend_i + 32 * u < n4) {
              float* du = d + u * (32 * 4 * kFHalf);
#pragma unroll
              for (int jj = 0; jj < 4; ++jj) {
                u32x4 o; o[0] = pc.stg[4 * u][jj]; o[1] = pc.stg[4 * u + 1][jj]; o[2] = pc.stg[4 * u + 2][jj]; o[3] = pc.stg[4 * u + 3][jj];
                *reinterpret_cast<u32x4*>(du + (jj ^ x_) * kFHalf) = o;
              }
            }
          }
          pc.pend = nullptr;
        }
        // (pair = position in the walk; lh = position of the half-slab in the walk, -1: nothing to load)
        int lh = -1, piece = 0, lbuf = 0;
        if (j >= 1 && j <= 4) { lh = 2 * pair + 2; piece = j - 1; lbuf = 0; }
        else if (j >= 5) { lh = 2 * pair + 3; piece = j - 5; lbuf = 1; }
#ifdef SD_PROFILING
        if (a.ablate & 4) lh = -1;
#endif
        if (LDSX && lh >= 0 && lh < nh && n4 > 0) {
          const float* src = G.xg + (long)((2 * pair_of(lh >> 1) + (lh & 1)) * kFHalf + cq * 4) * plane + G.wstart;
          pc.pend_i = piece * Q + lg;
          pc.pend = xs + lbuf * kFHalf * wstride + (4 * piece * Q) * kFHalf + cq * 4;
#pragma unroll
          for (int u = 0; u < kFStage / 4; ++u) {
            if (32 * u >= Q) break;
            int i = pc.pend_i + 32 * u;
            i = i < n4 ? i : n4 - 1;   // (past the end: loaded in vain, not stored)
#pragma unroll
            for (int c = 0; c < 4; ++c) pc.stg[4 * u + c] = *reinterpret_cast<const u32x4*>(src + (long)c * plane + 4 * i);
          }
        }
      };
      int pair = 0, j = 0;
      for (int s = 0; s < S;) {
        lstep(pa, pair, j);
        if (s + 1 < S) SD_FBAR();
        if (++j == 9) { j = 0; ++pair; }
        if (++s >= S) break;
        lstep(pb, pair, j);
        if (s + 1 < S) SD_FBAR();
        if (++j == 9) { j = 0; ++pair; }
        ++s;
      }
    }
  }
#ifdef SD_PROFILING
  if (a.dbg && lane == 0) {
    long long* d = a.dbg + ((long)blockIdx.x * 8 + wave) * 4;
    d[0] = __builtin_readcyclecounter() - p_begin; d[1] = p_wait; d[2] = p_setup; d[3] = wave;
  }
#endif
}

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

static int make_geom(DcnGeom& g, int N, int C, int H, int W, int kh, int kw, int pad_h, int pad_w,
                     int stride_h, int stride_w, int dil_h, int dil_w, int dgroup) {
  SD_REQUIRE(N >= 0 && C > 0 && H > 0 && W > 0, "bad input dimensions");
  SD_REQUIRE(kh > 0 && kw > 0 && stride_h > 0 && stride_w > 0 && dil_h > 0 && dil_w > 0,
             "bad kernel/stride/dilate");
  SD_REQUIRE(pad_h >= 0 && pad_w >= 0, "negative pad");
  SD_REQUIRE(dgroup > 0 && C % dgroup == 0, "input num_filter must divide deformable group size");
  g = DcnGeom{N, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, dgroup, 0, 0};
  g.Ho = (H + 2 * pad_h - (dil_h * (kh - 1) + 1)) / stride_h + 1;
  g.Wo = (W + 2 * pad_w - (dil_w * (kw - 1) + 1)) / stride_w + 1;
  SD_REQUIRE(g.Ho > 0 && g.Wo > 0, "kernel size exceed input");
  SD_REQUIRE((long)C * kh * kw * g.Ho * g.Wo < (1L << 31), "col matrix of one image >= 2^31 elements");
  SD_REQUIRE(dgroup * kh * kw * 2 <= 65535 && N <= 65535, "grid dimension too large");
  return SD_OK;
}

}  // namespace sd

using namespace sd;

extern "C" int sd_deform_im2col(const float* x, const float* offset, float* col, int N, int C,
                                int H, int W, int kh, int kw, int pad_h, int pad_w, int stride_h,
                                int stride_w, int dil_h, int dil_w, int dgroup, void* stream) {
  DcnGeom g;
  if (int e = make_geom(g, N, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, dgroup))
    return e;
  if (N == 0) return SD_OK;
  SD_REQUIRE(x && offset && col, "null tensor pointer");
  const int P = g.Ho * g.Wo;
  const size_t lds = ((size_t)H * W + W + 8) * sizeof(float);
  if (kh * kw <= kDcnMaxTaps && lds <= 64 * 1024 && (long)H * W < (1L << 28) &&
      tuning("dcn_im2col", 1) == 1) {
    constexpr int T = 256;
    const int vec = (((H * W) % 4 == 0 && ((uintptr_t)x & 15) == 0) ? 1 : 0) |
                    (tuning("dcn_window", 1) ? 0 : 2);
    // bit 0: non-temporal col stores (the product's setting); bits 1-2 switch parts off, profiling build only
    const int nt = 1 | (SD_PROF_TUNING("dcn_im2col_nt", 1) & 6);
    // (channel splits per (image, group, pixel tile): 2-8 measured in round 3, no gain -- one)
    const int nsplit = 1;
    if (kh * kw == 9)
      hipLaunchKernelGGL((deform_im2col_lds_kernel<T, 9>), dim3(cdiv(P, T), dgroup * nsplit, N),
                         dim3(T), lds, (hipStream_t)stream, x, offset, col, g, nsplit, vec, nt);
    else
      hipLaunchKernelGGL((deform_im2col_lds_kernel<T, 0>), dim3(cdiv(P, T), dgroup * nsplit, N),
                         dim3(T), lds, (hipStream_t)stream, x, offset, col, g, nsplit, vec, nt);
  } else {
    hipLaunchKernelGGL(deform_im2col_kernel, dim3(cdiv(P, 256), dgroup * kh * kw, N), dim3(256), 0,
                       (hipStream_t)stream, x, offset, col, g);
  }
  SD_LAUNCH_CHECK();
  return SD_OK;
}

// cmax / wsum (device; both or neither): bound of |col| and room for N * dgroup floats -- with them the
// four-channel kernel sums in fixed point (deform_col2im_chunk_kernel<.., true>)
static int col2im_impl(const float* col, const float* offset, float* dx, int req, int N, int C, int H, int W,
                       int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w,
                       int dgroup, void* stream, const unsigned* cmax, unsigned* wsum) {
  DcnGeom g;
  if (int e = make_geom(g, N, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, dgroup))
    return e;
  SD_REQUIRE(req == SD_REQ_NULL || req == SD_REQ_WRITE || req == SD_REQ_ADD, "bad req %d", req);
  if (N == 0 || req == SD_REQ_NULL) return SD_OK;
  SD_REQUIRE(col && offset && dx, "null tensor pointer");
  hipStream_t st = (hipStream_t)stream;
  {
    // four channels of a group per workgroup (the sample geometry is worked out once for them):
    // bands of at most 72 KB for the four planes, two workgroups of 512 lanes per CU
    constexpr int CC = 4, T = 512;
    const int P = g.Ho * g.Wo;
    const long budget4 = 72 * 1024;
    int nb4 = (int)(((long)CC * H * W * 4 + budget4 - 1) / budget4);
    const int rows4 = (H + nb4 - 1) / nb4;
    nb4 = (H + rows4 - 1) / rows4;
    if (tuning("dcn_col2im", 1) == 1 && (C / dgroup) % CC == 0 && P % 4 == 0 &&
        (((uintptr_t)col | (uintptr_t)offset) & 15) == 0 && (long)CC * rows4 * W * 4 <= 150 * 1024 &&
        nb4 <= 65535) {
      const size_t lds4 = (size_t)CC * rows4 * W * sizeof(float);
      const size_t ldsw = (size_t)H * W * sizeof(float);
      // weights as multiples of 2^-wshift: K2 * P of them (every sample of an image on one pixel) stay below 2^32
      int wshift = 20;
      while (wshift > 0 && (double)kh * kw * P * (double)(1u << wshift) >= 4294967296.0) --wshift;
      const bool fx = cmax && wsum && ldsw <= 150 * 1024 && kh * kw <= 65535 && wshift >= 8 &&
                      tuning("dcn_col2im_fx", 1) == 1;
      if (fx) {
        SD_HIP_CHECK(hipMemsetAsync(wsum, 0, sizeof(unsigned) * (size_t)N * dgroup, st));
        if (ldsw > 64 * 1024)
          SD_HIP_CHECK(hipFuncSetAttribute((const void*)deform_col2im_wsum_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsw));
        hipLaunchKernelGGL(deform_col2im_wsum_kernel, dim3(kh * kw, dgroup, N), dim3(512), ldsw, st, offset, wsum, g,
                           wshift);
        if (lds4 > 64 * 1024)
          SD_HIP_CHECK(hipFuncSetAttribute((const void*)deform_col2im_chunk_kernel<CC, T, true>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
        hipLaunchKernelGGL((deform_col2im_chunk_kernel<CC, T, true>), dim3(C / CC, nb4, N), dim3(T), lds4, st,
                           col, offset, dx, g, rows4, req == SD_REQ_ADD ? 1 : 0, cmax, wsum, wshift);
        SD_LAUNCH_CHECK();
        return SD_OK;
      }
      if (lds4 > 64 * 1024)
        SD_HIP_CHECK(hipFuncSetAttribute((const void*)deform_col2im_chunk_kernel<CC, T, false>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
      hipLaunchKernelGGL((deform_col2im_chunk_kernel<CC, T, false>), dim3(C / CC, nb4, N), dim3(T), lds4, st,
                         col, offset, dx, g, rows4, req == SD_REQ_ADD ? 1 : 0, nullptr, nullptr, 0);
      SD_LAUNCH_CHECK();
      return SD_OK;
    }
  }
  // row bands of at most 36 KB so that four workgroups share a CU
  const long budget = 36 * 1024;
  int nb = (int)(((long)H * W * 4 + budget - 1) / budget);
  if (nb < 1) nb = 1;
  int rows = (H + nb - 1) / nb;
  nb = (H + rows - 1) / rows;
  const size_t lds = (size_t)rows * W * sizeof(float);
  SD_REQUIRE(lds <= 150 * 1024, "DeformableConvolution: feature row of %d floats too wide", W);
  SD_REQUIRE(nb <= 65535, "too many row bands");
  if (lds > 64 * 1024)
    SD_HIP_CHECK(hipFuncSetAttribute((const void*)deform_col2im_kernel,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(deform_col2im_kernel, dim3(C, nb, N), dim3(256), lds, st, col, offset, dx, g,
                     rows, req == SD_REQ_ADD ? 1 : 0);
  SD_LAUNCH_CHECK();
  return SD_OK;
}

extern "C" int sd_deform_col2im(const float* col, const float* offset, float* dx, int req, int N,
                                int C, int H, int W, int kh, int kw, int pad_h, int pad_w,
                                int stride_h, int stride_w, int dil_h, int dil_w, int dgroup,
                                void* stream) {
  return col2im_impl(col, offset, dx, req, N, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w,
                     dgroup, stream, nullptr, nullptr);
}

extern "C" int sd_deform_col2im_coord(const float* col, const float* x, const float* offset,
                                      float* d_offset, int req, int N, int C, int H, int W, int kh,
                                      int kw, int pad_h, int pad_w, int stride_h, int stride_w,
                                      int dil_h, int dil_w, int dgroup, void* stream) {
  DcnGeom g;
  if (int e = make_geom(g, N, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w, dgroup))
    return e;
  SD_REQUIRE(req == SD_REQ_NULL || req == SD_REQ_WRITE || req == SD_REQ_ADD, "bad req %d", req);
  if (N == 0 || req == SD_REQ_NULL) return SD_OK;
  SD_REQUIRE(col && x && offset && d_offset, "null tensor pointer");
  const int P = g.Ho * g.Wo;
  const size_t lds = ((size_t)H * W + W + 8) * sizeof(float);
  if (kh * kw <= kDcnMaxTaps && lds <= 64 * 1024 && (long)H * W < (1L << 28) &&
      tuning("dcn_coord", 1) == 1) {
    constexpr int T = 256;
    const int vec = (((H * W) % 4 == 0 && ((uintptr_t)x & 15) == 0) ? 1 : 0) |
                    (tuning("dcn_window", 1) ? 0 : 2);
    if (kh * kw == 9)
      hipLaunchKernelGGL((deform_col2im_coord_lds_kernel<T, 9>), dim3(cdiv(P, T), dgroup, N),
                         dim3(T), lds, (hipStream_t)stream, col, x, offset, d_offset, g,
                         req == SD_REQ_ADD ? 1 : 0, vec);
    else
      hipLaunchKernelGGL((deform_col2im_coord_lds_kernel<T, 0>), dim3(cdiv(P, T), dgroup, N),
                         dim3(T), lds, (hipStream_t)stream, col, x, offset, d_offset, g,
                         req == SD_REQ_ADD ? 1 : 0, vec);
  } else
    hipLaunchKernelGGL(deform_col2im_coord_kernel, dim3(cdiv(P, 256), dgroup * 2 * kh * kw, N),
                       dim3(256), 0, (hipStream_t)stream, col, x, offset, d_offset, g,
                       req == SD_REQ_ADD ? 1 : 0);
  SD_LAUNCH_CHECK();
  return SD_OK;
}

static int gemm_f32_impl(int transA, int transB, int M, int N, int K, const float* A, int lda,
                         long strideA, const float* B, int ldb, long strideB, float* C, int ldc,
                         long strideC, int batch, int accumulate, const unsigned* amax, void* stream,
                         unsigned* cmax = nullptr) {
  SD_REQUIRE(M >= 0 && N >= 0 && K >= 0 && batch >= 0, "negative dimension");
  SD_REQUIRE(accumulate >= 0 && accumulate <= 2, "accumulate must be 0, 1 or 2");
  if (M == 0 || N == 0 || batch == 0) return SD_OK;
  SD_REQUIRE(A && B && C, "null matrix pointer");
  GemmArgs g{};
  g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K;
  // row-major: op(A) is M x K.  transA: A stored K x M
  g.sam = transA ? 1 : lda; g.sak = transA ? lda : 1;
  g.sbk = transB ? 1 : ldb; g.sbn = transB ? ldb : 1;
  g.ldc = ldc; g.strideA = strideA; g.strideB = strideB; g.strideC = strideC;
  g.mode = accumulate;
  g.amax = amax;
  g.cmax = cmax;
  if (K == 0) {
    if (accumulate == 0)
      for (int b = 0; b < batch; ++b)
        SD_HIP_CHECK(hipMemset2DAsync(C + (long)b * strideC, sizeof(float) * (size_t)ldc, 0,
                                      sizeof(float) * (size_t)N, (size_t)M, (hipStream_t)stream));
    return SD_OK;
  }
  return launch_gemm(g, batch, (hipStream_t)stream);
}

extern "C" int sd_gemm_f32(int transA, int transB, int M, int N, int K, const float* A, int lda,
                           long strideA, const float* B, int ldb, long strideB, float* C, int ldc,
                           long strideC, int batch, int accumulate, void* stream) {
  return gemm_f32_impl(transA, transB, M, N, K, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, batch,
                       accumulate, nullptr, stream);
}

extern "C" size_t sd_gemm_f32_workspace_bytes(void) { return 64; }

extern "C" int sd_gemm_f32_ws(int transA, int transB, int M, int N, int K, const float* A, int lda,
                              long strideA, const float* B, int ldb, long strideB, float* C, int ldc,
                              long strideC, int batch, int accumulate, void* workspace,
                              size_t workspace_bytes, void* stream) {
  unsigned* amax = nullptr;
  if (workspace && workspace_bytes >= 32 && M > 0 && N > 0 && K > 0 && batch > 0 && A && B &&
      tuning("deform_gemm_split", 2) == 2) {
    amax = reinterpret_cast<unsigned*>(((uintptr_t)workspace + 15) & ~(uintptr_t)15);
    hipStream_t st = (hipStream_t)stream;
    SD_HIP_CHECK(hipMemsetAsync(amax, 0, 8, st));
    // storage of op(A) (M x K): rows x cols = transA ? K x M : M x K, row stride lda; B likewise.
    // A batch stride of 0 is one shared matrix.
    launch_absmax(absmax_seg(A, transA ? K : M, transA ? M : K, lda, strideA, strideA == 0 ? 1 : batch, amax),
                  absmax_seg(B, transB ? N : K, transB ? K : N, ldb, strideB, strideB == 0 ? 1 : batch, amax + 1),
                  AbsSeg{}, st);
    SD_LAUNCH_CHECK();
  }
  return gemm_f32_impl(transA, transB, M, N, K, A, lda, strideA, B, ldb, strideB, C, ldc, strideC, batch,
                       accumulate, amax, stream);
}

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
static int deform_conv_fwd_impl(const float* x, const float* offset, const float* weight, const float* bias,
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

// ---- forward without a col matrix (fused sampling + GEMM) ----------------------------------------
static bool dcn_fused_shape_ok(int C, int H, int W, int kh, int kw, int dgroup) {
  return kh == 3 && kw == 3 && dgroup > 0 && C % dgroup == 0 && (C / dgroup) % 16 == 0 && ((long)H * W) % 4 == 0 &&
         (long)H * W < (1L << 28) && W + 16 < kFXFloats / 16 && tuning("dcn_fused", 1) == 1;
}

extern "C" size_t sd_deform_conv_fwd_nocol_workspace_bytes(int N, int C, int H, int W, int F, int kh, int kw,
                                                           int pad, int stride, int dil, int dgroup) {
  if (N <= 0 || C <= 0 || F <= 0) return 256;
  if (!dcn_fused_shape_ok(C, H, W, kh, kw, dgroup))
    return sd_deform_conv_workspace_bytes(N, C, H, W, kh, kw, pad, stride, dil);
  const long Ho = (H + 2 * pad - (dil * (kh - 1) + 1)) / stride + 1;
  const long Wo = (W + 2 * pad - (dil * (kw - 1) + 1)) / stride + 1;
  if (Ho <= 0 || Wo <= 0) return 256;   // (the call itself fails in make_geom: "kernel size exceed input")
  const size_t mtiles = (F + 255) / 256, nslab = C / 16;
  const size_t P = (size_t)Ho * (size_t)Wo;
  // the pre-split weights + the operand maxima + one flag per tile (at most one tile per pixel)
  return mtiles * nslab * 9 * 1024 * sizeof(uint4) + 512 + mtiles * (size_t)N * P * sizeof(int);
}

static int deform_conv_fwd_nocol_impl(const float* x, const float* offset, const float* weight, const float* bias,
                                      float* y, int N, int C, int H, int W, int F, int kh, int kw, int pad,
                                      int stride, int dil, int dgroup, void* workspace, size_t workspace_bytes,
                                      void* stream) {
  DcnGeom g;
  if (int e = make_geom(g, N, C, H, W, kh, kw, pad, pad, stride, stride, dil, dil, dgroup)) return e;
  SD_REQUIRE(F > 0, "num_filter must be positive");
  if (N == 0) return SD_OK;
  SD_REQUIRE(x && offset && weight && y, "null tensor pointer");
  if (!dcn_fused_shape_ok(C, H, W, kh, kw, dgroup))
    return deform_conv_fwd_impl(x, offset, weight, bias, y, N, C, H, W, F, kh, kw, pad, stride, dil, dgroup, 1,
                                workspace, workspace_bytes, stream);
  const size_t need = sd_deform_conv_fwd_nocol_workspace_bytes(N, C, H, W, F, kh, kw, pad, stride, dil, dgroup);
  if (!workspace || workspace_bytes < need)
    return fail(SD_ERR_WORKSPACE, "DeformableConvolution (fused forward) workspace too small: %zu < %zu bytes",
                workspace_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  const int mtiles = (F + 255) / 256, nslab = C / 16, P = g.Ho * g.Wo;
  uint4* apre = reinterpret_cast<uint4*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  const size_t apre_words = (size_t)mtiles * nslab * 9 * 1024;
  unsigned* amax = reinterpret_cast<unsigned*>(apre + apre_words);   // {max|W|, max|x|}
  SD_HIP_CHECK(hipMemsetAsync(amax, 0, 16, st));
  launch_absmax(absmax_seg(weight, 1, F * C * 9, F * C * 9, 0, 1, amax),
                absmax_seg(x, (long)N * C, H * W, H * W, 0, 1, amax + 1), AbsSeg{}, st);
  {
    const long total = (long)apre_words / 2;   // one thread per (hi, lo) pair
    hipLaunchKernelGGL(dcn_prep_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, weight,
                       apre, F, C, mtiles, nslab, amax);
  }
  DcnFusedArgs a{};
  a.x = x; a.offset = offset; a.apre = apre; a.bias = bias; a.y = y; a.g = g; a.F = F; a.mtiles = mtiles;
  a.nslab = nslab;
  a.amax = amax;
  a.x_aligned = ((uintptr_t)x & 15) == 0;
  // tiles per image: enough for <= 96 pixels each, then as many more as keeps the launch at the same
  // whole number of rounds of the CUs (one workgroup per CU): equal tiles instead of a ragged last round
  int T = cdiv(P, kFN);
  const long total0 = (long)N * T * mtiles;
  const long rounds = (total0 + kNumCU - 1) / kNumCU;
  const long fit = rounds * kNumCU / ((long)N * mtiles);
  if (fit > T) T = (int)(fit < P ? fit : P);
  a.tile_w = cdiv(P, T);
  const int tw = tuning("dcn_fused_tile", 0);
  if (tw >= 1 && tw <= kFN) a.tile_w = tw;
  a.tiles_per_image = cdiv(P, a.tile_w);
  SD_REQUIRE((long)a.tiles_per_image * 8 * cdiv(N, 8) * mtiles < (1L << 31), "too many tiles");
  a.flags = reinterpret_cast<int*>(amax + 64);   // (behind the maxima: 256 bytes into the 512 of slack)
  a.ablate = SD_PROF_TUNING("dcn_fused_ablate", 0);
  a.dbg = nullptr;
#ifdef SD_PROFILING
  a.dbg = reinterpret_cast<long long*>(((uintptr_t)(unsigned)SD_PROF_TUNING("roi_align_dbg_hi", 0) << 32) |
                                       (uintptr_t)(unsigned)SD_PROF_TUNING("roi_align_dbg_lo", 0));
#endif
  // (every call: the attribute is per device, and a process may drive several)
  SD_HIP_CHECK(hipFuncSetAttribute((const void*)dcn_fwd_fused_kernel<true>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, kFSmemBytes));
  SD_HIP_CHECK(hipFuncSetAttribute((const void*)dcn_fwd_fused_kernel<false>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, kFSmemBytes));
  const dim3 grid((unsigned)(a.tiles_per_image * 8 * cdiv(N, 8) * mtiles));
  hipLaunchKernelGGL(dcn_fwd_fused_kernel<true>, grid, dim3(kFThreads), kFSmemBytes, st, a);
  // tiles whose windows did not fit LDS (wild offsets) flagged themselves: the global-gather instance
  // redoes exactly those (every other block returns at once)
  hipLaunchKernelGGL(dcn_fwd_fused_kernel<false>, grid, dim3(kFThreads), kFSmemBytes, st, a);
  SD_LAUNCH_CHECK();
  return SD_OK;
}

extern "C" int sd_deform_conv_fwd_nocol(const float* x, const float* offset, const float* weight, float* y,
                                        int N, int C, int H, int W, int F, int kh, int kw, int pad,
                                        int stride, int dil, int dgroup, void* workspace,
                                        size_t workspace_bytes, void* stream) {
  return deform_conv_fwd_nocol_impl(x, offset, weight, nullptr, y, N, C, H, W, F, kh, kw, pad, stride, dil, dgroup,
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
  // fixed-point col2im: max|dcol| out of the GEMM's epilogue (amax[4 .. 4 + kCmaxSlots)) and N * dgroup
  // weight-sum bounds behind them, all inside the slack behind the col matrix (what is left of its 512
  // bytes depends on how far the caller's pointer was from a 256-byte boundary)
  unsigned* cmax = nullptr;
  unsigned* wsum = nullptr;
  const long slack_words = ((const char*)workspace + workspace_bytes - (const char*)amax) / 4;
  if (4 + kCmaxSlots + (long)N * dgroup <= slack_words && tuning("deform_gemm_split", 2) >= 1) {
    cmax = amax + 4;                // kCmaxSlots words
    wsum = amax + 4 + kCmaxSlots;   // N * dgroup words
  }
  SD_HIP_CHECK(hipMemsetAsync(amax, 0, cmax ? 16 + 4 * kCmaxSlots : 16, st));
  launch_absmax(absmax_seg(weight, 1, F * Kg, F * Kg, 0, 1, amax),
                absmax_seg(out_grad, (long)N * F, P, P, 0, 1, amax + 1),
                absmax_seg(x, (long)N * C, H * W, H * W, 0, 1, amax + 2), st);
  if (req_x != SD_REQ_NULL || req_offset != SD_REQ_NULL) {
    // dcol[n][grp] (K/G x P) = W[grp]^T (K/G x F/G) . dY[n][grp] (F/G x P)
    for (int q = 0; q < num_group; ++q)
      if (int e = gemm_f32_impl(1, 0, Kg, P, Fg, weight + (long)q * Fg * Kg, Kg, 0, out_grad + (long)q * Fg * P, P,
                                (long)F * P, col + (long)q * Kg * P, P, (long)K * P, N, 0, amax, stream, cmax))
        return e;
    if (int e = sd_deform_col2im_coord(col, x, offset, d_offset, req_offset, N, C, H, W, kh, kw, pad,
                                       pad, stride, stride, dil, dil, dgroup, stream))
      return e;
    if (int e = col2im_impl(col, offset, d_x, req_x, N, C, H, W, kh, kw, pad, pad, stride, stride, dil, dil,
                            dgroup, stream, cmax, wsum))
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
