// DeformableConvolution (v1) for gfx950: deformable im2col / col2im / col2im_coord.
//   upstream MXNet 1.6.0 src/operator/contrib/nn/deformable_im2col.cuh (un-vendored; oracle/deform_conv.c
//   restates it, "parity unpinned").  Layout: x (N,C,H,W); offset (N, dgroup*2*kh*kw, Ho, Wo) with channel
//   2*(i*kw+j) = dh, +1 = dw; col (N, C*kh*kw, Ho*Wo) with row (c*kh + i)*kw + j.
#include "deform_common.h"

namespace sd {

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
// on what a pixel can collect: max|col| of the workgroup's own values (round 6: taken optimistically from its
// first trip and verified behind the scatter, like roi_align_bwd_packed4 -- until round 5 the producing GEMM's
// epilogue had to record a global maximum, so only the layer's backward could use this path) times
// the largest sum of bilinear weights landing on one pixel, which depends on (image, group) only and
// is bounded per tap by deform_col2im_wsum_kernel (sum over the taps of each tap's largest pile-up).
// scale = the power of two that puts twice that bound below 2^29; one unit is then <= 2^-27 of the largest
// possible sum, and the result does not depend on the order of the adds.  Non-finite values, and a dynamic
// range the unit is too coarse for (kFxRangeBits), keep the compare-and-swap adds, which send inf / nan
// where the reference sends them.


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
                                                                const unsigned* __restrict__ wsum, int wshift) {
  extern __shared__ __attribute__((aligned(16))) float plane[];
  __shared__ unsigned s_ctl[4];        // [0] max|col| bits of the first trip, [1] of everything, [2] non-finite flag
  __shared__ int s_exp[2];             // [0] range margins | sample count of the sampled values (dynamic-range verdict)
  const int P = g.Ho * g.Wo, K2 = g.kh * g.kw;
  const int c0 = blockIdx.x * CC, n = blockIdx.z;
  const int row0 = blockIdx.y * band_rows, row1 = iminr(row0 + band_rows, g.H);
  const int band_elems = (row1 - row0) * g.W;
  const int tid = threadIdx.x;
  for (int i = tid; i < CC * band_elems; i += T) plane[i] = 0.f;   // (0.f and 0 are the same bits)
  if (tid < 4) s_ctl[tid] = 0u;
  if (tid < 2) s_exp[tid] = 0;
  const int cpg = g.C / g.dgroup, grp = c0 / cpg;
  const float* off = offset + ((long)n * g.dgroup + grp) * 2 * K2 * P;
  const float* cp = col + ((long)n * g.C + c0) * K2 * P;
  const long cstride = (long)K2 * P;  // col elements per channel
  // maxima as bit patterns of |col| (absbits4: a NaN survives); anything above FLT_MAX's pattern is non-finite
  auto wave_max_to = [&](unsigned m, int slot) {
    m = wave_max_u32(m);
    if ((tid & (kWave - 1)) == 0) {
      atomicMax(&s_ctl[slot], m);
      if (m > kFltMaxBits) atomicOr(&s_ctl[2], 1u);
    }
  };
  // Fixed point (round 6: self-contained -- no maximum from the producer of col, so the stand-alone
  // sd_deform_col2im_ws gets it too): the unit comes from max|col| of the workgroup's OWN 4 x K2 x P values x
  // the weight bound of its (image, group).  The maximum is taken optimistically from the values of every
  // thread's first trip (tap 0) with a factor two of headroom and verified behind the scatter, together with
  // the dynamic range (kFxRangeBits): a late outlier, a non-finite value or a range the unit is too coarse
  // for clear the planes and sum them again -- with the exact maximum, or with the fp32 compare-and-swap adds.
  bool fx = false;
  float scale = 1.f, weff = 0.f, gmax_used = 0.f;
  __syncthreads();   // planes and s_ctl are zero
  if (FX) {
    unsigned m0 = 0u;
    if (tid * 4 < P) {
#pragma unroll
      for (int cc = 0; cc < CC; ++cc) m0 = umaxr(m0, absbits4(*reinterpret_cast<const float4*>(cp + cc * cstride + tid * 4)));
    }
    wave_max_to(m0, 0);
    __syncthreads();
    gmax_used = __uint_as_float(s_ctl[0]);
    // (the integer weight sum is exact in a float up to 2^24 units; beyond that it is rounded to nearest:
    // one more unit of margin)
    weff = ((float)(wsum[(long)n * g.dgroup + grp] + 1u) / (float)(1u << wshift)) * 1.000001f;
  }
  // scale * 2 * (largest possible sum) < 2^29: the true maximum may be twice the optimistic one, and both sums of
  // a channel pair stay below 2^29 in magnitude (rounding of the individual adds and the slack of the fp32
  // weight sums stay far inside the remaining two bits)
  auto set_scale = [&](float gmax) {
    const float bound = gmax * weff;
    const unsigned bb = __float_as_uint(bound);
    const int e = (int)((bb >> 23) & 255);
    fx = false;
    scale = 1.f;
    if (bound == 0.f) {
      fx = true;   // nothing but zeros can arrive
    } else if (e != 255 && e != 0) {
      int es = 127 + 27 - (e - 127);   // scale = 2^(27 - floor(log2 bound))
      es = es > 254 ? 254 : es;
      if (es >= 1) {
        scale = __uint_as_float((unsigned)es << 23);
        fx = true;
      }
    }
  };
  if (FX && !s_ctl[2]) set_scale(gmax_used);
  unsigned m_all = 0u;
  // (the unit here is 2^-28 .. 2^-27 of the bound where the RoIAlign planes have 2^-30 .. 2^-29: two bits less range;
  // sampled: channel 0 of tap 0, the first 16 K pixels -- < 4096 values)
  int e_acc = 0;
  const int e_thr = fx_range_thr(gmax_used, (int)ceilf(fminr(weff, 1e9f))) + 2;
  for (int attempt = 0; attempt < 2; ++attempt) {
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
        if (FX && attempt == 0) {
#pragma unroll
          for (int cc = 0; cc < CC; ++cc) {
            m_all = umaxr(m_all, absbits4(cv[cc]));
            if (cc == 0 && tap == 0 && p4 < 16000) e_acc += fx_range_sample(cv[cc].x, e_thr);
          }
        }
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
    if (!FX || !fx || attempt > 0) break;
    // was the optimistic maximum enough, and is the unit fine enough for what was streamed?
    wave_max_to(m_all, 1);
    {
      const int es = wave_sum_i32(e_acc);
      if ((tid & (kWave - 1)) == 0) atomicAdd(&s_exp[0], es);   // integer sums: the order of the waves does not matter
    }
    __syncthreads();
    const float gmax_true = __uint_as_float(s_ctl[1]);
    const bool fine = fx_range_fine(s_exp[0], gmax_used, gmax_true);
    if (!s_ctl[2] && fine && gmax_true <= 2.f * gmax_used) break;   // also when every value is zero
    __syncthreads();   // every thread has read the verdict before the planes are cleared
    for (int i = tid; i < CC * band_elems; i += T) plane[i] = 0.f;
    if (s_ctl[2] || !fine) fx = false;
    else set_scale(gmax_true);
    __syncthreads();
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

// The same kernel as a software pipeline around vmcnt (round 5; 3x3 taps, 16-byte windows).  vmcnt counts loads
// and stores in order: in the loop above the wait for a channel's window loads is also a wait for the previous
// channel's nine col stores per lane (2.3 KB per wave on their way to HBM), on every one of the 64 channels, and
// the loads themselves sit between two barriers with nothing to hide them.  Here the NEXT channel's window loads
// go out before this channel's values are computed and stored, and are waited for behind the stores with
// exactly those in flight (s_waitcnt vmcnt(3): every trip issues the same memory operations -- every wave
// writes three rows, lanes past the last pixel repeat the last pixel's work, the last trip reloads its own
// window): the load latency hides behind the sampling, the stores are never waited for.  To make room for the two window words per lane the four
// bilinear weights of a tap are kept as their two fractions and multiplied out per channel (the same products in
// the same order: the same bits).
template <int T>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(5, 8)))
void deform_im2col_pipe_kernel(const float* __restrict__ x, const float* __restrict__ offset,
                               float* __restrict__ col, DcnGeom g) {
  extern __shared__ __attribute__((aligned(16))) float xs[];
  constexpr int K2 = 9;
  const int P = g.Ho * g.Wo, plane = g.H * g.W;
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  const int p0 = blockIdx.x * T;
  const int p = iminr(p0 + tid, P - 1);   // (lanes past the last pixel: the last pixel again)
  const int grp = blockIdx.y, n = blockIdx.z;
  const int cpg = g.C / g.dgroup;
  // The col values leave as WHOLE tile rows: a tap's T values are T * 4 contiguous bytes of col, staged in LDS
  // ([tap][pixel]) and written behind the barrier that ends the sampling by ONE wave as 16 bytes per lane -- 1 KB
  // pieces instead of four 256-byte ones (written as 256-byte pieces the 620 MB move at 2.7 TB/s, whatever the
  // bytes per lane: `profiles/r05u_*`).  Every wave writes exactly three rows (T / 64 = 4 waves, 9 taps: wave w
  // writes taps 2 w and 2 w + 1 and its own quarter of tap 8), so the count of stores per trip stays static.
  static_assert(T == 256, "store schedule: four waves, nine taps");
  float* stg = xs + ((plane + g.W + 8 + 3) & ~3);   // [K2][T], 16-byte aligned for the ds_read_b128 row reads whatever W is
  int info[kDcnMaxTaps];
  float lh[K2], lw[K2];
  {
    const int h_col = p / g.Wo, w_col = p % g.Wo;
    const int h_in = h_col * g.stride_h - g.pad_h, w_in = w_col * g.stride_w - g.pad_w;
    const float* off = offset + ((long)n * g.dgroup + grp) * 2 * K2 * P + p;
    float oh[K2], ow[K2];
#pragma unroll
    for (int tap = 0; tap < K2; ++tap) {  // all offset loads in flight together
      oh[tap] = off[(long)(2 * tap) * P];
      ow[tap] = off[(long)(2 * tap + 1) * P];
    }
#pragma unroll
    for (int tap = 0; tap < K2; ++tap) {
      const Sample s = im2col_sample(g, h_in, w_in, tap / 3, tap % 3, oh[tap], ow[tap]);
      info[tap] = dcn_pack(s.ok, s.h_low, s.w_low, s.h_high, s.w_high, g.W);
      lh[tap] = s.lh; lw[tap] = s.lw;
      __builtin_amdgcn_sched_barrier(0);  // one tap's temporaries at a time (register pressure)
    }
  }
  int wstart, wcount;
  dcn_window(info, g.W, plane, 1, reinterpret_cast<int*>(xs + plane + g.W + 4), tid, wstart, wcount);
  const int n4 = wcount >> 2, last4 = n4 > 0 ? n4 - 1 : 0;
  const long ch0 = (long)n * g.C + (long)grp * cpg;
  float4* d4 = reinterpret_cast<float4*>(xs);
  auto window_to_lds = [&](const float4& r0, const float4& r1, const float4* s4) {
    if (tid < n4) d4[tid] = r0;
    if (tid + T < n4) d4[tid + T] = r1;
    for (int i = tid + 2 * T; i < n4; i += T) d4[i] = s4[i];   // windows beyond 2 T x 16 bytes (wide offsets)
  };
  {
    const float4* s4 = reinterpret_cast<const float4*>(x + ch0 * plane + wstart);
    const float4 r0 = s4[iminr(tid, last4)], r1 = s4[iminr(tid + T, last4)];
    window_to_lds(r0, r1, s4);
  }
  typedef float f4v __attribute__((ext_vector_type(4)));
  auto store_rows = [&](float* out) {   // three 16-byte stores per lane; pieces past the last pixel are masked (P % 4 == 0)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int tap = 2 * wave + k;
      if (p0 + 4 * lane < P)
        __builtin_nontemporal_store(*reinterpret_cast<const f4v*>(stg + tap * T + 4 * lane),
                                    reinterpret_cast<f4v*>(out + (tap * P + p0 + 4 * lane)));
    }
    if (lane < 16 && p0 + wave * kWave + 4 * lane < P)
      __builtin_nontemporal_store(*reinterpret_cast<const f4v*>(stg + 8 * T + wave * kWave + 4 * lane),
                                  reinterpret_cast<f4v*>(out + (8 * P + p0 + wave * kWave + 4 * lane)));
  };
  for (int c = 0; c < cpg; ++c) {
    const long ch = ch0 + c;
    __syncthreads();  // this channel's window is in LDS, the previous channel's rows have left the staging area
    // the NEXT channel's window loads go out first (the last trip reloads its own: same instruction count)
    const float4* s4 = reinterpret_cast<const float4*>(x + (ch0 + (c + 1 < cpg ? c + 1 : c)) * plane + wstart);
    float4 r0 = s4[iminr(tid, last4)], r1 = s4[iminr(tid + T, last4)];
#pragma unroll
    for (int tap = 0; tap < K2; ++tap) {
      asm volatile("" : "+v"(info[tap]), "+v"(lh[tap]), "+v"(lw[tap]));  // keep the unpacking and the products inside the loop (registers)
      const int in = info[tap];
      const Corners q = dcn_corners(xs, in, g.W);
      const float hh = 1 - lh[tap], hw = 1 - lw[tap];
      const float w1 = hh * hw, w2 = hh * lw[tap], w3 = lh[tap] * hw, w4 = lh[tap] * lw[tap];
      float v = (w1 * q.x1 + w2 * q.x2 + w3 * q.x3 + w4 * q.x4);
      if (!(in & kDcnInside)) v = 0.f;
      stg[tap * T + tid] = v;
    }
    __syncthreads();  // everybody has read this channel's window and staged its values
    store_rows(col + ch * K2 * P);   // wave-uniform base + 32-bit lane offset
    // the window loads, with this channel's three stores still in flight: s_waitcnt vmcnt(3)
    asm volatile("" : "+v"(r0.x), "+v"(r0.y), "+v"(r0.z), "+v"(r0.w), "+v"(r1.x), "+v"(r1.y), "+v"(r1.z), "+v"(r1.w));
    window_to_lds(r0, r1, s4);
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

int make_geom(DcnGeom& g, int N, int C, int H, int W, int kh, int kw, int pad_h, int pad_w,
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
    if (kh * kw == 9 && kw == 3 && vec == 1 && nt == 1 && C / dgroup >= 2 && P % 4 == 0 && ((uintptr_t)col & 15) == 0 &&
        lds + 16 + 9 * T * sizeof(float) <= 31 * 1024 && tuning("dcn_im2col", 1) == 1 &&
        tuning("dcn_im2col_pipe", 1) == 1)
      hipLaunchKernelGGL((deform_im2col_pipe_kernel<T>), dim3(cdiv(P, T), dgroup, N), dim3(T),
                         lds + 16 + 9 * T * sizeof(float), (hipStream_t)stream, x, offset, col, g);
    else if (kh * kw == 9)
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

// wsum (device, N * dgroup words, or null): room for the per-(image, group) weight bounds -- with it the
// four-channel kernel sums in fixed point (deform_col2im_chunk_kernel<.., true>)
int sd::col2im_impl(const float* col, const float* offset, float* dx, int req, int N, int C, int H, int W,
                       int kh, int kw, int pad_h, int pad_w, int stride_h, int stride_w, int dil_h, int dil_w,
                       int dgroup, void* stream, unsigned* wsum) {
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
      const bool fx = wsum && ldsw <= 150 * 1024 && kh * kw <= 65535 && wshift >= 8 &&
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
                           col, offset, dx, g, rows4, req == SD_REQ_ADD ? 1 : 0, wsum, wshift);
        SD_LAUNCH_CHECK();
        return SD_OK;
      }
      if (lds4 > 64 * 1024)
        SD_HIP_CHECK(hipFuncSetAttribute((const void*)deform_col2im_chunk_kernel<CC, T, false>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
      hipLaunchKernelGGL((deform_col2im_chunk_kernel<CC, T, false>), dim3(C / CC, nb4, N), dim3(T), lds4, st,
                         col, offset, dx, g, rows4, req == SD_REQ_ADD ? 1 : 0, nullptr, 0);
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
                     dgroup, stream, nullptr);
}

extern "C" size_t sd_deform_col2im_workspace_bytes(int N, int dgroup) {
  if (N <= 0 || dgroup <= 0) return 16;
  return (size_t)N * dgroup * sizeof(unsigned) + 16;
}

extern "C" int sd_deform_col2im_ws(const float* col, const float* offset, float* dx, int req, int N,
                                   int C, int H, int W, int kh, int kw, int pad_h, int pad_w,
                                   int stride_h, int stride_w, int dil_h, int dil_w, int dgroup,
                                   void* workspace, size_t workspace_bytes, void* stream) {
  unsigned* wsum = nullptr;
  if (workspace && N > 0 && dgroup > 0) {
    wsum = reinterpret_cast<unsigned*>(((uintptr_t)workspace + 3) & ~(uintptr_t)3);
    if ((const char*)(wsum + (size_t)N * dgroup) > (const char*)workspace + workspace_bytes)
      return fail(SD_ERR_WORKSPACE, "deformable col2im workspace too small: %zu < %zu bytes", workspace_bytes,
                  sd_deform_col2im_workspace_bytes(N, dgroup));
  }
  return col2im_impl(col, offset, dx, req, N, C, H, W, kh, kw, pad_h, pad_w, stride_h, stride_w, dil_h, dil_w,
                     dgroup, stream, wsum);
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
