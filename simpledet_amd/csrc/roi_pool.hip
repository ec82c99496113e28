// ROIPooling_v1 forward/backward for gfx950.
//   reference: operator_cxx/roi_pooling_v1.cu:48-113 (forward), :115-152 (backward scatter),
//              roi_pooling_v1.cc:39-126 (CPU forward, same result), roi_pooling_v1-inl.h:70-133
//              (pre-fill, req handling).
// forward (roi_pool_fwd_plane_kernel): workgroup = (channel, image) with the channel plane in LDS,
//   see the kernel (0.65 -> 0.30 ms on the C4 shape).  roi_pool_fwd = 0 selects the
//   wave-per-(roi, channel) kernel (lane = output bin, one uncoalesced global load per visited
//   pixel), also the fallback for planes over 64 KB.
// backward (roi_pool_bwd_lds_kernel): workgroup = (image, channel, row band) with the band of dX
//   in LDS; the bins of the image's RoIs whose arg-max falls in the band are added with an LDS
//   compare-and-swap and the band is written once: no zero-fill pass, no global atomics
//   (1.71 -> 0.24 ms on the C4 shape); with C % 4 == 0 four channels per workgroup and 16-byte
//   loads (roi_pool_bwd_lds4_kernel: 0.11 ms).  roi_pool_bwd = 2 keeps one channel per workgroup,
//   0 selects the reference structure (zero-fill + global atomics), also the fallback when the RoI
//   list does not fit in LDS.
#include "common.h"
#include "../../include/simpledet_ops.h"
#include <float.h>
#include <math.h>

namespace sd {

struct PoolArgs {
  const float* data;
  const float* rois;
  float* out;
  float* maxidx;
  int B, C, H, W, K, PH, PW;
  float scale;
  int ablate;  // (profiling build: 1 no scan, 2 no output stores)
};

__global__ __launch_bounds__(256) void roi_pool_fwd_kernel(PoolArgs a) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int waves_per_block = blockDim.x / kWave;
  const int PP = a.PH * a.PW;
  const long nwork = (long)a.K * a.C;  // (roi, channel) pairs
  for (long wk = (long)blockIdx.x * waves_per_block + wave; wk < nwork;
       wk += (long)gridDim.x * waves_per_block) {
    const int n = (int)(wk / a.C), c = (int)(wk % a.C);
    const float* r = a.rois + (long)n * 5;
    const int roi_batch_ind = (int)r[0];
    // round(): half away from zero, on the float product (roi_pooling_v1.cu:70-73)
    const int roi_start_w = (int)roundf(r[1] * a.scale);
    const int roi_start_h = (int)roundf(r[2] * a.scale);
    const int roi_end_w = (int)roundf(r[3] * a.scale);
    const int roi_end_h = (int)roundf(r[4] * a.scale);
    const int roi_width = imaxr(roi_end_w - roi_start_w + 1, 1);
    const int roi_height = imaxr(roi_end_h - roi_start_h + 1, 1);
    const float bin_size_h = (float)roi_height / (float)a.PH;
    const float bin_size_w = (float)roi_width / (float)a.PW;
    const bool batch_ok = roi_batch_ind >= 0 && roi_batch_ind < a.B;
    const float* plane = a.data + ((long)(batch_ok ? roi_batch_ind : 0) * a.C + c) * a.H * a.W;
    const long obase = wk * PP;
    for (int bin = lane; bin < PP; bin += kWave) {
      const int ph = bin / a.PW, pw = bin % a.PW;
      int hstart = (int)floorf((float)ph * bin_size_h);
      int wstart = (int)floorf((float)pw * bin_size_w);
      int hend = (int)ceilf((float)(ph + 1) * bin_size_h);
      int wend = (int)ceilf((float)(pw + 1) * bin_size_w);
      hstart = iminr(imaxr(hstart + roi_start_h, 0), a.H);
      hend = iminr(imaxr(hend + roi_start_h, 0), a.H);
      wstart = iminr(imaxr(wstart + roi_start_w, 0), a.W);
      wend = iminr(imaxr(wend + roi_start_w, 0), a.W);
      const bool is_empty = (hend <= hstart) || (wend <= wstart) || !batch_ok;
      float maxval = is_empty ? 0.f : -FLT_MAX;
      int maxi = -1;
      if (!is_empty) {
        for (int h = hstart; h < hend; ++h) {
          const float* row = plane + (long)h * a.W;
          for (int w = wstart; w < wend; ++w) {
            const float v = row[w];
            if (v > maxval) {
              maxval = v;
              maxi = h * a.W + w;
            }
          }
        }
      }
      a.out[obase + bin] = maxval;
      a.maxidx[obase + bin] = (float)maxi;
    }
  }
}

// Plane-resident forward: workgroup = (channel, image).  The channel plane (H*W floats, 16.8 KB on
// the C4 shape) is read ONCE, coalesced, into LDS; the image's RoIs are listed in chunks of 256
// with their bin boundaries per axis (PH + PW packed (start, end) pairs: 14 instead of 4 x 49
// floor/ceil/clamp evaluations per RoI) and the largest bin extent per axis; then wave <- RoI,
// lane <- bin: every visited pixel is one ds_read_b32 instead of one uncoalesced global load
// (435 M visits for 1024 RoIs x 1024 channels, ~16 clocks of the texture-address unit per wave
// load before).  The scan loops run to the RoI's largest bin extent -- wave-uniform trip counts,
// loop control on the scalar unit -- and a lane whose bin is smaller re-reads its last row /
// column: an equal value never wins the strict comparison, so every bin still sees the
// reference's scan order (rows, then columns, strict >) and gives identical values and arg-max.
// (The first version ran per-lane loops: 95% VALU-busy, 11 VALU per LDS read.)
constexpr int kPoolChunk = 256;

// CC = 4 (round 3): the workgroup holds FOUR consecutive channel planes, interleaved per pixel
// ([pixel][4]), so one ds_read_b128 serves four channels and the clamped column / row index of a
// visit is computed once for them (7 -> ~4 VALU per channel visit); the four 196-byte output rows
// of a RoI are one contiguous 784-byte run.
// Its outputs leave as 16-byte stores: a wave stages the 4 x 49 values of a RoI in LDS and 49 lanes
// write the 784-byte run as dwordx4 (4-byte stores cost ~6x the dwordx4 time per byte, and the two
// outputs are 92 % of this op's traffic).  One workgroup of 1024 lanes per CU (the four planes, the
// RoI tables and the staging rows take 110 KB).
template <int PHc, int PWc, int CC, int T>  // PHc = PWc = 0: runtime pooled size
__global__ __launch_bounds__(T) void roi_pool_fwd_plane_kernel(PoolArgs a) {
  constexpr int CHUNK = kPoolChunk, NWAVE = T / kWave;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int HW = a.H * a.W;
  const int PH = PHc ? PHc : a.PH, PW = PWc ? PWc : a.PW, PP = PH * PW;
  const int ES = 4 + PH + PW;  // table words per RoI: index, valid, max extents, row / column pairs
  float* plane = smem;
  int* tab = reinterpret_cast<int*>(smem + ((HW * CC + 3) & ~3));
  constexpr bool STAGE = CC == 4 && PHc * PWc > 0 && PHc * PWc <= kWave;
  // per wave 2 x (CC x PP) floats, 16-byte aligned, behind the tables
  float* stage = reinterpret_cast<float*>(tab + ((CHUNK * ES + 3) & ~3)) +
                 (STAGE ? (threadIdx.x / kWave) * 2 * ((CC * PHc * PWc + 3) & ~3) : 0);
  // chunk counters, one per chunk parity: the counter of the NEXT chunk is reset after this
  // chunk's listing barrier, when no wave can still be reading it (a single counter reset at the
  // top of the loop raced with slow waves reading the previous chunk's count)
  __shared__ int cnt[2];
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  // Workgroup i runs on XCD i % 8, and the 196-byte output rows of neighbouring channels share
  // cache lines: neighbouring channels go to the SAME XCD so that one L2 assembles whole lines
  // (channel = XCD's eighth of the range + position within it).
  const int b = blockIdx.y;
  const int G = a.C / CC;  // channel groups
  const int c = CC * ((G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x);
  const float* src = a.data + ((long)b * a.C + c) * HW;
  if (CC == 1) {
    for (int i = tid; i < HW; i += T) plane[i] = src[i];
  } else {
#pragma unroll
    for (int cc = 0; cc < CC; ++cc)
      for (int i = tid; i < HW; i += T) plane[i * CC + cc] = src[(long)cc * HW + i];
  }

  if (tid == 0) cnt[0] = 0;
  for (int k0 = 0, par = 0; k0 < a.K; k0 += CHUNK, par ^= 1) {
    __syncthreads();  // (also: plane complete, previous chunk's items done)
    // two threads per RoI: the first lists it and does the rows, the second the columns
    const int k = k0 + (tid >> 1), axis = tid & 1;
    if (k < a.K && (tid >> 1) < CHUNK) {
      const float* r = a.rois + (long)k * 5;
      const int ind = (int)r[0];
      const bool batch_ok = ind >= 0 && ind < a.B;
      // a RoI whose batch index names no image pools nothing; image 0's workgroups write its zeros
      const bool mine = batch_ok ? ind == b : b == 0;
      int slot = 0;
      if (mine && axis == 0) slot = atomicAdd(&cnt[par], 1);
      slot = __shfl(slot, (tid & 63) & ~1);
      if (mine) {
        int* e = tab + ES * slot;
        // round(): half away from zero, on the float product (roi_pooling_v1.cu:70-73)
        const int start = (int)roundf(r[1 + (axis ^ 1)] * a.scale);  // axis 0: rows (y1), 1: columns (x1)
        const int end = (int)roundf(r[3 + (axis ^ 1)] * a.scale);
        const int len = imaxr(end - start + 1, 1);
        const int P = axis ? PW : PH, size = axis ? a.W : a.H;
        const float bin_size = (float)len / (float)P;
        int* dst = e + 4 + (axis ? PH : 0);
        int ext = 0;
        for (int p = 0; p < P; ++p) {
          int lo = (int)floorf((float)p * bin_size);
          int hi = (int)ceilf((float)(p + 1) * bin_size);
          lo = iminr(imaxr(lo + start, 0), size);
          hi = iminr(imaxr(hi + start, 0), size);
          dst[p] = lo | (hi << 16);
          ext = imaxr(ext, hi - lo);
        }
        e[2 + axis] = ext;
        if (axis == 0) {
          e[0] = k;
          e[1] = batch_ok;
        }
      }
    }
    __syncthreads();
    const int n = cnt[par];
    if (tid == 0) cnt[par ^ 1] = 0;
    for (int ri = wave; ri < n; ri += NWAVE) {
      const int* e = tab + ES * ri;
      const long obase = ((long)__builtin_amdgcn_readfirstlane(e[0]) * a.C + c) * PP;
      const bool valid = __builtin_amdgcn_readfirstlane(e[1]) != 0;
      const int hext = SD_ABLATE(a, 1) ? 0 : __builtin_amdgcn_readfirstlane(e[2]), wext = __builtin_amdgcn_readfirstlane(e[3]);
      for (int bin0 = 0; bin0 < PP; bin0 += kWave) {
        const int bin = bin0 + lane;
        const bool active = bin < PP;
        const int bb = active ? bin : 0;
        const int ph = bb / PW, pw = bb - ph * PW;
        const int hb = e[4 + ph], wb = e[4 + PH + pw];
        int hstart = hb & 0xffff, hend = hb >> 16, wstart = wb & 0xffff, wend = wb >> 16;
        const bool is_empty = (hend <= hstart) || (wend <= wstart) || !valid;
        if (is_empty) {  // scans pixel (0, 0), result discarded
          hstart = wstart = 0;
          hend = wend = 1;
        }
        const int hl = hend - 1, wl = wend - 1;
        float maxval[CC];
        int maxi[CC];
#pragma unroll
        for (int cc = 0; cc < CC; ++cc) {
          maxval[cc] = -FLT_MAX;
          maxi[cc] = -1;
        }
        for (int dh = 0; dh < hext; ++dh) {
          const int ib = iminr(hstart + dh, hl) * a.W;
          const float* row = plane + ib * CC;
#pragma unroll 4
          for (int dw = 0; dw < wext; ++dw) {
            const int w = iminr(wstart + dw, wl);
            float v[CC];
            if (CC == 4) {
              const float4 q = *reinterpret_cast<const float4*>(row + w * 4);
              v[0] = q.x; v[1 % CC] = q.y; v[2 % CC] = q.z; v[3 % CC] = q.w;
            } else {
#pragma unroll
              for (int cc = 0; cc < CC; ++cc) v[cc] = row[w * CC + cc];
            }
#pragma unroll
            for (int cc = 0; cc < CC; ++cc)
              if (v[cc] > maxval[cc]) {
                maxval[cc] = v[cc];
                maxi[cc] = ib + w;
              }
          }
        }
        if (STAGE) {
          constexpr int RUN = (CC * PHc * PWc + 3) & ~3;
          if (active) {
#pragma unroll
            for (int cc = 0; cc < CC; ++cc) {
              stage[cc * PP + bin] = is_empty ? 0.f : maxval[cc];
              stage[RUN + cc * PP + bin] = (float)(is_empty ? -1 : maxi[cc]);
            }
          }
          // (the LDS operations of one wave complete in order: no barrier between the two halves)
          if (lane * 4 < CC * PP) {
            const float4 o = *reinterpret_cast<const float4*>(stage + lane * 4);
            const float4 m = *reinterpret_cast<const float4*>(stage + RUN + lane * 4);
            if (!(SD_ABLATE(a, 2))) {
              *reinterpret_cast<float4*>(a.out + obase + lane * 4) = o;
              *reinterpret_cast<float4*>(a.maxidx + obase + lane * 4) = m;
            } else if (o.x == 12345.678f && m.x == 4.5f) {
              a.out[0] = 0.f;
            }
          }
        } else if (active) {
#pragma unroll
          for (int cc = 0; cc < CC; ++cc) {
            a.out[obase + cc * PP + bin] = is_empty ? 0.f : maxval[cc];
            a.maxidx[obase + cc * PP + bin] = (float)(is_empty ? -1 : maxi[cc]);
          }
        }
      }
    }
  }
}

struct PoolBwdArgs {
  const float* dy;
  const float* rois;
  const float* maxidx;
  float* dx;
  int B, C, H, W, K, PP;
};

__global__ __launch_bounds__(256) void roi_pool_bwd_kernel(PoolBwdArgs a) {
  const long count = (long)a.K * a.C * a.PP;
  for (long index = (long)blockIdx.x * blockDim.x + threadIdx.x; index < count;
       index += (long)gridDim.x * blockDim.x) {
    const int c = (int)((index / a.PP) % a.C);
    const int n = (int)(index / a.PP / a.C);
    const int argmax = (int)a.maxidx[index];
    if (argmax != -1) {
      const int b = (int)a.rois[(long)n * 5];
      if (b >= 0 && b < a.B && argmax >= 0 && argmax < a.H * a.W)
        atomicAdd(a.dx + ((long)b * a.C + c) * a.H * a.W + argmax, a.dy[index]);
    }
  }
}


// ------------------------------------------------------------------------------------------------
// LDS-plane backward
// ------------------------------------------------------------------------------------------------

// grid: x = channel, y = row band, z = image
__global__ __launch_bounds__(512) void roi_pool_bwd_lds_kernel(PoolBwdArgs a, int band_rows,
                                                               int req_add) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int T = 512;
  const int tid = threadIdx.x;
  const int c = blockIdx.x, b = blockIdx.z;
  const int row0 = blockIdx.y * band_rows, row1 = iminr(row0 + band_rows, a.H);
  const int band_elems = (row1 - row0) * a.W;
  const int plane_pad = (band_elems + 3) & ~3;
  float* plane = smem;
  int* list = reinterpret_cast<int*>(smem + plane_pad);
  int* nlist = list + a.K;
  for (int i = tid; i < plane_pad; i += T) plane[i] = 0.f;
  if (tid == 0) *nlist = 0;
  __syncthreads();
  for (int n = tid; n < a.K; n += T)
    if ((int)a.rois[(long)n * 5] == b) list[atomicAdd(nlist, 1)] = n;
  __syncthreads();
  const int nitems = *nlist * a.PP;
  const int lo = row0 * a.W, hi = row1 * a.W;
  for (int it = tid; it < nitems; it += T) {
    const int n = list[it / a.PP], bin = it % a.PP;
    const long idx = ((long)n * a.C + c) * a.PP + bin;
    const int argmax = (int)a.maxidx[idx];
    if (argmax >= lo && argmax < hi) lds_add_cas(plane + (argmax - lo), a.dy[idx]);
  }
  __syncthreads();
  float* dst = a.dx + (((long)b * a.C + c) * a.H + row0) * a.W;
  for (int i = tid; i < band_elems; i += T) dst[i] = req_add ? dst[i] + plane[i] : plane[i];
}

// Four channels per workgroup (round 3): the arg-max and gradient rows of (RoI, c..c+3) are 784
// contiguous, 16-byte aligned bytes, read as dwordx4 (a quarter of the load instructions, no
// division per element); four whole planes of dX in LDS (67 KB at 50x84, two workgroups per CU).
//   grid: x = channel quad, y = image
__global__ __launch_bounds__(512) void roi_pool_bwd_lds4_kernel(PoolBwdArgs a, int req_add) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int T = 512, CC = 4;
  const int tid = threadIdx.x;
  const int G = a.C / CC;
  const int c = CC * ((G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x);
  const int b = blockIdx.y;
  const int HW = a.H * a.W;
  float* plane = smem;  // [CC][HW]
  int* list = reinterpret_cast<int*>(smem + CC * HW);
  int* nlist = list + a.K;
  for (int i = tid; i < CC * HW; i += T) plane[i] = 0.f;
  if (tid == 0) *nlist = 0;
  __syncthreads();
  for (int n = tid; n < a.K; n += T)
    if ((int)a.rois[(long)n * 5] == b) list[atomicAdd(nlist, 1)] = n;
  __syncthreads();
  const int PP = a.PP;           // CC * PP is a multiple of 4 (host)
  const int per = CC * PP / 4;   // float4 units per RoI
  const int nunits = *nlist * per;
  for (int u = tid; u < nunits; u += T) {
    const int r = u / per, L = u - r * per;
    const long idx = ((long)list[r] * a.C + c) * PP + 4 * L;
    const float4 am = *reinterpret_cast<const float4*>(a.maxidx + idx);
    const float4 g = *reinterpret_cast<const float4*>(a.dy + idx);
    const float amv[4] = {am.x, am.y, am.z, am.w}, gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int e = 4 * L + k;
      const int cc = (e >= PP) + (e >= 2 * PP) + (e >= 3 * PP);
      const int argmax = (int)amv[k];
      if (argmax >= 0 && argmax < HW) lds_add_cas(plane + cc * HW + argmax, gv[k]);
    }
  }
  __syncthreads();
  float* dst = a.dx + ((long)b * a.C + c) * HW;  // the four planes are contiguous
  for (int i = tid; i < CC * HW; i += T) dst[i] = req_add ? dst[i] + plane[i] : plane[i];
}

}  // namespace sd

using namespace sd;

extern "C" int sd_roi_pool_v1_fwd(const float* data, const float* rois, float* out, float* maxidx,
                                  int B, int C, int H, int W, int K, int pooled_h, int pooled_w,
                                  float spatial_scale, void* stream) {
  SD_REQUIRE(B >= 0 && C >= 0 && K >= 0 && H > 0 && W > 0, "bad dimensions");
  SD_REQUIRE(pooled_h > 0 && pooled_w > 0, "pooled_size must be nonzero");
  SD_REQUIRE((long)K * C * pooled_h * pooled_w < (1L << 31), "output has >= 2^31 elements");
  SD_REQUIRE((long)H * W < (1L << 24), "plane too large for a float argmax index");
  if ((long)K * C == 0) return SD_OK;
  SD_REQUIRE(data && rois && out && maxidx, "null tensor pointer");
  PoolArgs a{data, rois, out, maxidx, B, C, H, W, K, pooled_h, pooled_w, spatial_scale, SD_PROF_TUNING("roi_pool_fwd_ablate", 0)};
  const size_t lds = (size_t)((((long)H * W + 3) & ~3L) + kPoolChunk * (4 + pooled_h + pooled_w)) * 4;
  // four planes + tables + per-wave staging rows (7x7 only), one workgroup of 1024 lanes per CU
  const size_t lds4 = (size_t)((long)H * W * 4 + ((kPoolChunk * (4 + 7 + 7) + 3) & ~3) + 16 * 2 * 196) * 4;
  const int mode = tuning("roi_pool_fwd", 1);  // 1 default (four channels per workgroup when they fit), 2 one channel, 0 per-item kernel
  if (lds <= 64 * 1024 && H <= 32767 && W <= 32767 && B >= 1 && B <= 65535 && mode != 0) {
    // (with B == 0 every batch index is out of range: the wave-per-item kernel writes the zeros)
    hipStream_t st = (hipStream_t)stream;
    if (mode == 1 && C % 4 == 0 && pooled_h == 7 && pooled_w == 7 && lds4 <= 150 * 1024 &&
        (((uintptr_t)out | (uintptr_t)maxidx) & 15) == 0) {
      SD_HIP_CHECK(hipFuncSetAttribute((const void*)roi_pool_fwd_plane_kernel<7, 7, 4, 1024>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
      hipLaunchKernelGGL((roi_pool_fwd_plane_kernel<7, 7, 4, 1024>), dim3(C / 4, B), dim3(1024), lds4, st, a);
    } else if (pooled_h == 7 && pooled_w == 7) {
      hipLaunchKernelGGL((roi_pool_fwd_plane_kernel<7, 7, 1, 512>), dim3(C, B), dim3(512), lds, st, a);
    } else {
      hipLaunchKernelGGL((roi_pool_fwd_plane_kernel<0, 0, 1, 512>), dim3(C, B), dim3(512), lds, st, a);
    }
    SD_LAUNCH_CHECK();
    return SD_OK;
  }
  const long nwork = (long)K * C;
  const int grid = (int)((nwork + 3) / 4 < kNumCU * 16 ? (nwork + 3) / 4 : kNumCU * 16);
  hipLaunchKernelGGL(roi_pool_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  SD_LAUNCH_CHECK();
  return SD_OK;
}

extern "C" int sd_roi_pool_v1_bwd(const float* out_grad, const float* rois, const float* maxidx,
                                  float* d_data, float* d_rois, int req_data, int req_rois, int B,
                                  int C, int H, int W, int K, int pooled_h, int pooled_w,
                                  float spatial_scale, void* stream) {
  (void)spatial_scale;
  SD_REQUIRE(B >= 0 && C >= 0 && K >= 0 && H > 0 && W > 0, "bad dimensions");
  SD_REQUIRE(pooled_h > 0 && pooled_w > 0, "pooled_size must be nonzero");
  SD_REQUIRE(req_data == SD_REQ_NULL || req_data == SD_REQ_WRITE || req_data == SD_REQ_ADD,
             "ROIPooling: Backward doesn't support kWriteInplace.");
  SD_REQUIRE(req_rois == SD_REQ_NULL || req_rois == SD_REQ_WRITE || req_rois == SD_REQ_ADD,
             "ROIPooling: Backward doesn't support kWriteInplace.");
  hipStream_t st = (hipStream_t)stream;
  const size_t dx_bytes = (size_t)B * C * H * W * sizeof(float);
  if (req_data != SD_REQ_NULL && dx_bytes) {
    SD_REQUIRE(d_data, "d_data is null");
    const long count = (long)K * C * pooled_h * pooled_w;
    SD_REQUIRE(count == 0 || (out_grad && rois && maxidx), "null tensor pointer");
    PoolBwdArgs a{out_grad, rois, maxidx, d_data, B, C, H, W, K, pooled_h * pooled_w};
    // row bands of at most 36 KB so that four workgroups share a CU
    const long budget = 36 * 1024;
    int nb = (int)(((long)H * W * 4 + budget - 1) / budget);
    if (nb < 1) nb = 1;
    int rows = (H + nb - 1) / nb;
    nb = (H + rows - 1) / rows;
    const size_t lds = (size_t)((((long)rows * W + 3) & ~3L) * 4) + (size_t)(K + 4) * 4;
    const size_t lds4 = (size_t)H * W * 16 + (size_t)(K + 4) * 4;
    const int mode = tuning("roi_pool_bwd", 1);  // 1 default, 2 one channel per workgroup, 0 global atomics
    if (mode == 1 && C % 4 == 0 && lds4 <= 76 * 1024 && B <= 65535 &&
        (((uintptr_t)out_grad | (uintptr_t)maxidx) & 15) == 0) {
      if (lds4 > 64 * 1024)
        SD_HIP_CHECK(hipFuncSetAttribute((const void*)roi_pool_bwd_lds4_kernel,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds4));
      hipLaunchKernelGGL(roi_pool_bwd_lds4_kernel, dim3(C / 4, B), dim3(512), lds4, st, a,
                         req_data == SD_REQ_ADD ? 1 : 0);
      SD_LAUNCH_CHECK();
    } else if (lds <= 150 * 1024 && C <= 65535 * 32 && nb <= 65535 && B <= 65535 && mode != 0) {
      if (lds > 64 * 1024)
        SD_HIP_CHECK(hipFuncSetAttribute((const void*)roi_pool_bwd_lds_kernel,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(roi_pool_bwd_lds_kernel, dim3(C, nb, B), dim3(512), lds, st, a, rows,
                         req_data == SD_REQ_ADD ? 1 : 0);
      SD_LAUNCH_CHECK();
    } else {
      if (req_data == SD_REQ_WRITE) SD_HIP_CHECK(hipMemsetAsync(d_data, 0, dx_bytes, st));
      if (count) {
        const int grid = (int)((count + 255) / 256 < kNumCU * 32 ? (count + 255) / 256 : kNumCU * 32);
        hipLaunchKernelGGL(roi_pool_bwd_kernel, dim3(grid), dim3(256), 0, st, a);
        SD_LAUNCH_CHECK();
      }
    }
  }
  if (req_rois == SD_REQ_WRITE && K > 0) {  // roi_pooling_v1-inl.h:130-132
    SD_REQUIRE(d_rois, "d_rois is null but req_rois == write");
    SD_HIP_CHECK(hipMemsetAsync(d_rois, 0, (size_t)K * 5 * sizeof(float), st));
  }
  return SD_OK;
}
