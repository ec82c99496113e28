// Band list of the RoIAlign backward: shared by the rois-only pre-pass (roi_align_prep.hip) and by the
// workgroups of roi_align_bwd_packed4 that run without a workspace (roi_align_bwd.hip).
#pragma once
#include "roi_align_common.h"

namespace sd {

// RoIs of image `img` whose taps can fall on rows [row0, row1) of level `lvl`, in RoI order (ballot
// + prefix over the waves, no atomic slot counter: the list and everything derived from it are a
// deterministic function of the inputs), and the band's weight bound (see roi_align_bwd_packed4).
// list[0 .. count), nlist[0] = count, nlist[1] = bound; wcnt: THREADS / 64 words of scratch.
// Ends with a barrier.
template <int PH, int PW, int THREADS>
__device__ __forceinline__ void bwd_band_list(const BwdFusedArgs& a, int lvl, int img, int nbands, int row0,
                                              int row1, float4 rb0, int* list, int* nlist, int* wcnt) {
  constexpr int NW = THREADS / kWave;
  const int tid = threadIdx.x, wave = tid / kWave, lane = tid & (kWave - 1);
  const int H = a.L.H[lvl];
  const float scale = a.L.scale[lvl];
  int base = 0, bound_sum = 0;
  for (int r0 = 0; r0 < a.R; r0 += THREADS) {
    const int r = r0 + tid;
    bool take = r < a.R && !(SD_ABLATE(a, 4));  // (profiling build, 4: empty lists)
    int weight = 0;
    if (take) {
      const float4 rb = r0 == 0 ? rb0 : *reinterpret_cast<const float4*>(a.rois + ((long)img * a.R + r) * 4);
      if (a.filter) take = fpn_level(rb.x, rb.y, rb.z, rb.w, a.L) == lvl;
      if (take && nbands > 1) {
        // conservative row range of every tap of this RoI (taps lie within the clipped bins +-1)
        float s = fminr(fmaxr(rb.y * scale, 0.f), (float)(H - 1));
        float e = fminr(fmaxr(rb.w * scale, 0.f), (float)(H - 1));
        float lo = fminr(s, e) - 2.f, hi = fmaxr(s, e) + 2.f;
        if (hi < (float)row0 || lo > (float)(row1 - 1)) take = false;
      }
      if (take) {
        // bins of this RoI that can put weight on one pixel (see the header comment); a degenerate
        // or NaN width compares false and counts every bin
        // (hardware reciprocal, 1 ulp, with a 1e-5 safety factor: the count may only err upwards)
        const float bwx = (rb.z - rb.x) * scale * (1.f / (float)PW), bwy = (rb.w - rb.y) * scale * (1.f / (float)PH);
        const float fx = 2.00002f * __builtin_amdgcn_rcpf(bwx), fy = 2.00002f * __builtin_amdgcn_rcpf(bwy);
        const int nx = (bwx > 0.f && fx < (float)PW) ? iminr((int)fx + 2, PW) : PW;
        const int ny = (bwy > 0.f && fy < (float)PH) ? iminr((int)fy + 2, PH) : PH;
        weight = nx * ny;
      }
    }
    const unsigned long long mask = __ballot(take);
    bound_sum += wave_sum_i32(weight);
    if (r0 > 0) __syncthreads();  // wcnt of the previous sweep has been read
    if (lane == 0) wcnt[wave] = __popcll(mask);
    __syncthreads();
    int off = base, total = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      const int n = wcnt[w];
      if (w < wave) off += n;
      total += n;
    }
    if (take) list[off + __popcll(mask & ((1ull << lane) - 1))] = r;
    base += total;
  }
  if (lane == 0) atomicAdd(nlist + 1, bound_sum);
  if (tid == 0) nlist[0] = base;
  __syncthreads();
}

}  // namespace sd
