// RoIAlign, the rois-only pre-passes.  Everything here is a pure function of `rois` and the level
// geometry: the forward's item lists / tap entries / coordinate table (band_prep_block), the backward's
// band lists / tap tables (bwd_lists_block), and ONE launch that does both for a training step
// (roi_prep_merged_kernel).
#include "roi_align_lists.h"

namespace sd {

// ---- pre-pass ----
// blocks [0, B * nlvl): item lists of (level, image); then entries; then (packed) the coordinate table
template <int POOL>
__device__ __forceinline__ void band_prep_block(const BandArgs& A, const int pblock, int nlist, int nent) {
  const FwdArgs& a = A.f;
  const BandPlan& P = A.p;
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  constexpr int LPR = POOL <= 8 ? 8 : 16;            // lanes per RoI in the list pass
  constexpr int RPP = kBandThreads / LPR;             // RoIs per pass
  if (pblock < nlist) {
    // one workgroup per (level, image, quarter of the image's RoIs): every quarter writes its own
    // segment of every band's list, so no workgroup waits for another and a list is the
    // concatenation of kBandSub segments
    __shared__ int hist[kBandMaxBands], cursor[kBandMaxBands];
    const int sub = pblock % kBandSub, lvl = (pblock / kBandSub) % a.L.nlvl;
    const int img = pblock / (kBandSub * a.L.nlvl);
    if (a.L.stride[lvl] < 0) return;
    int first_valid = 0;
    while (a.L.stride[first_valid] < 0) ++first_valid;
    const int H = a.L.H[lvl], W = a.L.W[lvl], nb = P.nbands[lvl], owned = P.owned[lvl];
    const float scale = a.L.scale[lvl];
    const int rsub = (a.R + kBandSub - 1) / kBandSub;
    const int rbeg = sub * rsub, rend = rbeg + rsub < a.R ? rbeg + rsub : a.R;
    if (tid < kBandMaxBands) hist[tid] = 0;
    __syncthreads();
    const int rr = tid / LPR, p = tid % LPR;
    unsigned* items = P.items + (((long)img * SD_MAX_FPN_LEVELS + lvl) * kBandSub + sub) * rsub * POOL;
    // band of item (n, p): -1 none (not this level / idle lane), -2 the RoI goes to the exact path
    auto classify = [&](int n) -> int {
      int band = -1;
      bool bad = false, mine = false, emp_r = true, emp_c = true;
      int lv = -2;
      if (n < rend) {
        const float4 bx = *reinterpret_cast<const float4*>(a.rois + ((long)img * a.R + n) * 4);
        lv = a.L.nlvl > 1 ? fpn_level(bx.x, bx.y, bx.z, bx.w, a.L) : 0;
        mine = lv == lvl;
        if (mine && p < POOL) {
          float val[2], frac[2];
          int offr[4], offc[4];
          const int cr = axis_samples(p, POOL, bx.y, bx.w, scale, H, 1, val, frac, offr);
          const int cc = axis_samples(p, POOL, bx.x, bx.z, scale, W, 1, val, frac, offc);
          bad = cr >= 3 || cc >= 3;
          emp_r = cr < 0;
          emp_c = cc < 0;
          band = 0;
          if (cr >= 1) {
            const int first = offr[0], last = cr >= 2 ? offr[3] : offr[1];
            if (nb > 1 && last - first > P.halo[lvl]) bad = true;
            band = first / owned;
          }
        }
      }
      // RoI-wide verdict: any bad bin row / column sends the whole RoI to the exact path
      const unsigned long long bm = __ballot(bad);
      const int sh = (lane / LPR) * LPR;
      const unsigned long long rmask = (1ull << LPR) - 1;
      bool roi_bad = ((bm >> sh) & rmask) != 0;
      // a RoI that pools nothing anywhere (every bin row or every bin column empty: the zero boxes
      // fpn_roi_assign hands the per-level ops, padding rows) is constant output: flag 2, the
      // exact-path workgroups just store it
      const bool all_r = ((__ballot(!emp_r) >> sh) & rmask) == 0, all_c = ((__ballot(!emp_c) >> sh) & rmask) == 0;
      const bool roi_void = mine && (all_r || all_c);
      roi_bad = roi_bad || roi_void;
      if (n < rend && p == 0) {  // (rewritten with the same value when classify runs twice)
        if (mine) P.fbflag[(long)img * a.R + n] = roi_void ? 2 : (roi_bad ? 1 : 0);
        else if (lv < 0 && lvl == first_valid) P.fbflag[(long)img * a.R + n] = 2;
      }
      return roi_bad ? -2 : band;
    };
    constexpr int KP = 4;  // passes whose verdicts stay in registers between the two phases
    int keep[KP];
    const int npass = (rend - rbeg + RPP - 1) / RPP;
#pragma unroll
    for (int k = 0; k < KP; ++k) {
      keep[k] = -1;
      if (k < npass) {
        keep[k] = classify(rbeg + k * RPP + rr);
        if (keep[k] >= 0) atomicAdd(&hist[keep[k]], 1);
      }
    }
    for (int k = KP; k < npass; ++k) {
      const int band = classify(rbeg + k * RPP + rr);
      if (band >= 0) atomicAdd(&hist[band], 1);
    }
    __syncthreads();
    if (tid == 0) {
      int run = 0;
      for (int b = 0; b < nb; ++b) {
        cursor[b] = run;
        P.seg[(P.unit_base[lvl] + img * nb + b) * kBandSub + sub] = make_int2(run, hist[b]);
        run += hist[b];
      }
    }
    __syncthreads();
    // ordered inside a wave (a RoI's bin rows stay adjacent), waves reserve ranges atomically
    auto emit = [&](int n, int band) {
      unsigned long long todo = __ballot(band >= 0);
      while (todo) {
        const int src = __builtin_ctzll(todo);
        const int bb = __builtin_amdgcn_readlane(band, src);
        const unsigned long long m = __ballot(band == bb);
        int base = 0;
        if (lane == src) base = atomicAdd(&cursor[bb], __popcll(m));
        base = __builtin_amdgcn_readlane(base, src);
        if (band == bb)
          items[base + __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0))] =
              (unsigned)n | ((unsigned)p << 16);
        todo &= ~m;
      }
    };
#pragma unroll
    for (int k = 0; k < KP; ++k)
      if (k < npass) emit(rbeg + k * RPP + rr, keep[k]);
    for (int k = KP; k < npass; ++k) {
      const int n = rbeg + k * RPP + rr;
      emit(n, classify(n));
    }
    return;
  }
  if (pblock == nlist && tid < kBandMaxUnits) P.chan_ctr[tid] = 0;
  const int eb = pblock - nlist;
  const int nroi = a.B * a.R;
  if (eb < nent) {
    // ---- entries: one thread per (RoI, axis, bin) ----
    const long e = (long)eb * kBandThreads + tid;
    if (e >= (long)nroi * 2 * POOL) return;
    const int p = (int)(e % POOL), ax = (int)((e / POOL) % 2), n = (int)(e / (2 * POOL));
    const float4 bx = *reinterpret_cast<const float4*>(a.rois + (long)n * 4);
    const int lvl = a.L.nlvl > 1 ? fpn_level(bx.x, bx.y, bx.z, bx.w, a.L) : 0;
    if (lvl < 0) return;
    const int H = a.L.H[lvl], W = a.L.W[lvl];
    float val[2] = {0.f, 0.f}, frac[2] = {0.f, 0.f};
    int off[4];
    const int cnt = ax == 0 ? axis_samples(p, POOL, bx.y, bx.w, a.L.scale[lvl], H, W, val, frac, off)
                            : axis_samples(p, POOL, bx.x, bx.z, a.L.scale[lvl], W, 1, val, frac, off);
    const float qnan = __int_as_float(0x7fc00000);
    const bool v0 = cnt >= 1, v1 = cnt >= 2;
    uint4 en;
    if (ax == 0) {
      en.x = (v0 ? (unsigned)off[0] | (off[1] != off[0] ? 1u << 20 : 0u) : 0u) | (cnt < 0 ? 1u << 31 : 0u);
      en.y = v1 ? (unsigned)off[2] | (off[3] != off[2] ? 1u << 20 : 0u) : 0u;
    } else {
      en.x = (v0 ? (unsigned)off[0] | (off[1] == off[0] ? 1u << 12 : 0u) : 0u) |
             (v1 ? (unsigned)off[2] << 13 | (off[3] == off[2] ? 1u << 25 : 0u) : 0u) | (cnt < 0 ? 1u << 26 : 0u);
      en.y = 0u;
    }
    en.z = __float_as_uint(v0 ? frac[0] : qnan);
    en.w = __float_as_uint(v1 ? frac[1] : qnan);
    (ax == 0 ? P.rowent : P.colent)[(long)n * POOL + p] = en;
    if (P.rowval) (ax == 0 ? P.rowval : P.colval)[(long)n * POOL + p] = make_float2(val[0], val[1]);
    return;
  }
  // ---- packed arg-max: the per-RoI sample-coordinate table (what roi_coords_kernel writes) ----
  if (a.amax8) {
    const long e = (long)(eb - nent) * kBandThreads + tid;
    if (e >= (long)nroi * 6 * POOL) return;
    const int j = (int)(e % (6 * POOL)), n = (int)(e / (6 * POOL));
    const float4 bx = *reinterpret_cast<const float4*>(a.rois + (long)n * 4);
    const int lvl = a.L.nlvl > 1 ? fpn_level(bx.x, bx.y, bx.z, bx.w, a.L) : 0;
    if (lvl < 0) return;
    const bool row = j < 3 * POOL;
    const int jj = row ? j : j - 3 * POOL;
    const float v = row ? sample_coord(jj / 3, POOL, bx.y, bx.w, a.L.scale[lvl], a.L.H[lvl], jj % 3)
                        : sample_coord(jj / 3, POOL, bx.x, bx.z, a.L.scale[lvl], a.L.W[lvl], jj % 3);
    float* cb = a.coords + (long)n * kCoordWords * (POOL + POOL);
    cb[j] = v;
    store_tap(cb + 3 * (POOL + POOL) + 2 * j, v, row ? a.L.H[lvl] : a.L.W[lvl]);
  }
}

// The pre-pass is its own launch.  (Fusing it into the band kernel -- the first blocks build the
// tables and publish a per-launch tag, the band workgroups poll for it -- was built and measured:
// 168-190 us against 100; the agent-scope release / acquire traffic of a few hundred 1024-thread
// blocks costs far more than the ~10 us of a second launch.)
template <int POOL>
__global__ __launch_bounds__(kBandThreads) void roi_fwd_prep_kernel(BandArgs A) {
  band_prep_block<POOL>(A, (int)blockIdx.x, A.p.nlist, A.p.nent);
}

// Pre-pass per (level, image, band) unit: its list into the workspace, read by the 256 channel
// workgroups of roi_align_bwd_packed4 instead of being rebuilt by each of them.
// A pure function of `rois` (and the level geometry): the sample coordinates are recomputed with
// sample_coord() -- the very expression the forward fills its coordinate table with -- instead of
// being read from that table, so these blocks do not depend on the forward's pre-pass and can run
// in the SAME launch (roi_prep_merged_kernel below): one rois-only pre-pass per training step.
template <int PH, int PW, int THREADS, int PARTS>
__device__ __forceinline__ void bwd_lists_block(const BwdFusedArgs& a, int block, float* smem) {
  int* list = reinterpret_cast<int*>(smem);
  int* nlist = list + a.R;
  const int tid = threadIdx.x;
  // PARTS workgroups per unit: each builds the (cheap) list, the first stores it, all share
  // the tap entries -- the entry loop is a chain of dependent round trips (list -> box ->
  // entry), so more workgroups shorten the pre-pass
  // (round 6: one more workgroup per unit when the bands get a per-pixel weight bound -- it builds the list and
  // the bound and no tap entries, so the bound is off the tap workgroups' chain)
  const int NP = PARTS + (a.pix_bound_words > 0 ? 1 : 0);
  const int unit = block / NP, part = block % NP;
  int li = 0;
  while (li + 1 < a.nlaunch && unit >= a.unit_base[li + 1]) ++li;
  const int lvl = a.order[li];
  const int u = unit - a.unit_base[li];
  const int nbands = a.nbands[lvl];
  const int img = u / nbands, band = u % nbands;
  const int row0 = band * a.band_rows[lvl];
  const int row1 = iminr(row0 + a.band_rows[lvl], a.L.H[lvl]);
  float4 rb0 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tid < a.R) rb0 = *reinterpret_cast<const float4*>(a.rois + ((long)img * a.R + tid) * 4);
  if (tid < 8) nlist[tid] = 0;
  __syncthreads();
  bwd_band_list<PH, PW, THREADS>(a, lvl, img, nbands, row0, row1, rb0, list, nlist, nlist + 8);
  int* dst = a.ws_list + (long)unit * (a.R + 2);
  const int nl = nlist[0];
  // Round 6: the weight bound per CELL of 4 x 4 pixels.  bwd_band_list sums nx * ny over every RoI of the band
  // (1,000-1,800 at the baseline), but a RoI can put weight only on the pixels of its own clipped box +-2: every RoI
  // adds its nx * ny to the rectangle of cells its footprint touches in a 2-D difference array of the band, a row
  // prefix (wave scans) and a column prefix (only its maximum is kept) give the largest bound any cell -- hence any
  // pixel -- of the band can reach: typically 30-150.  The fixed-point unit of the channel workgroups is 2^-30 of
  // (max|dY| x this bound): an order of magnitude finer, and an order of magnitude more dynamic range before a
  // workgroup has to take the float adds (kFxRangeBits).  (Per pixel the array is 7 k words for a P2 band and its
  // scans put 2 us on the pre-pass, which the forward waits for; per cell it is 600 words and costs nothing
  // measurable.)  Done by the unit's extra workgroup (part == PARTS), once per unit and launch.
  const int H = a.L.H[lvl], W = a.L.W[lvl];
  constexpr int CS = 2;   // log2 of the cell edge
  const int DW = ((W - 1) >> CS) + 2, DH = ((row1 - row0 - 1) >> CS) + 2;
  const bool pix = DW * DH <= a.pix_bound_words;
  if (part == 0) {
    if (tid == 0) dst[0] = nl;
    if (tid == 1 && !pix) dst[1] = nlist[1];
    for (int i = tid; i < nl; i += THREADS) dst[2 + i] = list[i];
  }
  if (part == PARTS && !pix) return;   // this level's bands are too large for the difference array
  if (part == PARTS) {
    constexpr int NW = THREADS / kWave;
    const int wave = tid / kWave, lane = tid & (kWave - 1);
    int* D = nlist + 8 + 16;
    for (int i = tid; i < DW * DH; i += THREADS) D[i] = 0;
    __syncthreads();
    const float scale = a.L.scale[lvl];
    for (int j = tid; j < nl; j += THREADS) {
      const float4 rb = *reinterpret_cast<const float4*>(a.rois + ((long)img * a.R + list[j]) * 4);
      // pixels the taps of this RoI can reach: the clipped box +-2 (the list's own row test); anything else
      // (NaN / inf coordinates) counts on the whole band
      const float xs = fminr(fmaxr(rb.x * scale, 0.f), (float)(W - 1)), xe = fminr(fmaxr(rb.z * scale, 0.f), (float)(W - 1));
      const float ys = fminr(fmaxr(rb.y * scale, 0.f), (float)(H - 1)), ye = fminr(fmaxr(rb.w * scale, 0.f), (float)(H - 1));
      const float xlo = fminr(xs, xe) - 2.f, xhi = fmaxr(xs, xe) + 2.f;
      const float ylo = fminr(ys, ye) - 2.f, yhi = fmaxr(ys, ye) + 2.f;
      int x0 = 0, x1 = W - 1, y0 = 0, y1 = H - 1;
      if (xlo >= -2.f && xhi <= (float)(W + 1)) { x0 = imaxr((int)floorf(xlo), 0); x1 = iminr((int)ceilf(xhi), W - 1); }
      if (ylo >= -2.f && yhi <= (float)(H + 1)) { y0 = imaxr((int)floorf(ylo), 0); y1 = iminr((int)ceilf(yhi), H - 1); }
      y0 = imaxr(y0, row0) - row0;
      y1 = iminr(y1, row1 - 1) - row0;
      if (y0 > y1) continue;
      x0 >>= CS; x1 >>= CS; y0 >>= CS; y1 >>= CS;   // cells
      const float bwx = (rb.z - rb.x) * scale * (1.f / (float)PW), bwy = (rb.w - rb.y) * scale * (1.f / (float)PH);
      const float fx = 2.00002f * __builtin_amdgcn_rcpf(bwx), fy = 2.00002f * __builtin_amdgcn_rcpf(bwy);
      const int nx = (bwx > 0.f && fx < (float)PW) ? iminr((int)fx + 2, PW) : PW;
      const int ny = (bwy > 0.f && fy < (float)PH) ? iminr((int)fy + 2, PH) : PH;
      const int w = nx * ny;
      atomicAdd(D + y0 * DW + x0, w);
      atomicAdd(D + y0 * DW + x1 + 1, -w);
      atomicAdd(D + (y1 + 1) * DW + x0, -w);
      atomicAdd(D + (y1 + 1) * DW + x1 + 1, w);
    }
    __syncthreads();
    for (int y = wave; y < DH; y += NW) {  // inclusive prefix along the row, 64 columns per step
      int carry = 0;
      for (int xb = 0; xb < DW; xb += kWave) {
        const int x = xb + lane;
        const int v = wave_scan_i32(x < DW ? D[y * DW + x] : 0) + carry;
        if (x < DW) D[y * DW + x] = v;
        carry = __builtin_amdgcn_readlane(v, kWave - 1);
      }
    }
    __syncthreads();
    int m = 0;
    for (int x = tid; x < DW; x += THREADS) {  // prefix down the columns: only its maximum is kept
      int run = 0;
      for (int y = 0; y < DH; ++y) {
        run += D[y * DW + x];
        m = imaxr(m, run);
      }
    }
    if (m > 0) atomicMax(nlist + 5, m);
    __syncthreads();
    if (tid == 0) dst[1] = iminr(nlist[1], nlist[5]);
    return;
  }
  // ... and the tap entries of the listed RoIs (see roi_align_bwd_packed4): per sample coordinate
  // {neighbours, fraction} with the backward's own expressions, the row neighbours as offsets
  // inside this band (0xffff: outside), 8 bytes each, in list order
  if (a.ws_taps) {
    constexpr int NE = 3 * (PH + PW);
    const int H = a.L.H[lvl], W = a.L.W[lvl];
    const float scale = a.L.scale[lvl];
    float* tdst = a.ws_taps + (long)unit * a.R * (2 * NE);
    for (int i = part * THREADS + tid; i < nl * NE; i += PARTS * THREADS) {
      const int j = i / NE, e = i - j * NE;
      const bool row = e < 3 * PH;
      const int ee = row ? e : e - 3 * PH;
      const float4 bx = *reinterpret_cast<const float4*>(a.rois + ((long)img * a.R + list[j]) * 4);
      // == the forward's coords[roi][e] (band_prep_block / roi_coords_kernel), bit for bit
      const float v = row ? sample_coord(ee / 3, PH, bx.y, bx.w, scale, H, ee % 3)
                          : sample_coord(ee / 3, PW, bx.x, bx.z, scale, W, ee % 3);
      const int size = row ? H : W;
      const int lo = iminr(imaxr((int)floorf(v), 0), size - 1);
      const int hi = iminr(imaxr((int)ceilf(v), 0), size - 1);
      const float frac = (lo == hi) ? 0.5f : (v - (float)lo);  // (v - low) / (high - low), high - low == 1
      // (byte offsets into the band, so that a bin's LDS address is one add: 4 * col, 4 * (row - row0) * W < 65535)
      unsigned w0 = (unsigned)(4 * lo) | ((unsigned)(4 * hi) << 16);
      if (row) {
        const unsigned o0 = (lo >= row0 && lo < row1) ? (unsigned)(4 * (lo - row0) * W) : 0xffffu;
        const unsigned o1 = (hi >= row0 && hi < row1) ? (unsigned)(4 * (hi - row0) * W) : 0xffffu;
        w0 = o0 | (o1 << 16);
      }
      *reinterpret_cast<float2*>(tdst + (long)j * (2 * NE) + 2 * e) = make_float2(__uint_as_float(w0), frac);
    }
  }
}

template <int PH, int PW>
__global__ __launch_bounds__(512) void roi_align_bwd_lists(BwdFusedArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  bwd_lists_block<PH, PW, 512, kListSplit>(a, (int)blockIdx.x, smem);
}

// ONE rois-only pre-pass for a training step: the forward's item lists / tap entries / coordinate
// table (band_prep_block) and the backward's band lists / tap tables (bwd_lists_block) in a single
// launch -- both are pure functions of `rois` (VERDICT r3 "Next 3(i)").  Blocks [0, nfwd) do the
// forward's part, the rest the backward's (two 1024-thread blocks per backward unit).
template <int POOL>
__global__ __launch_bounds__(kBandThreads) void roi_prep_merged_kernel(BandArgs A, BwdFusedArgs b, int nfwd) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  if ((int)blockIdx.x < nfwd) band_prep_block<POOL>(A, (int)blockIdx.x, A.p.nlist, A.p.nent);
  else bwd_lists_block<POOL, POOL, kBandThreads, kMergedListSplit>(b, (int)blockIdx.x - nfwd, smem);
}


int launch_fwd_prep(const BandArgs& A, int pool, int nblocks, const BwdFusedArgs* bplan, hipStream_t st) {
  const FwdArgs& a = A.f;
  if (bplan) {
    const dim3 g((unsigned)(nblocks + bplan->lists_units * (kMergedListSplit + (bplan->pix_bound_words > 0 ? 1 : 0))));
    const size_t lds = (size_t)(a.R + 8 + 16 + bplan->pix_bound_words) * 4;
    if (pool == 7) hipLaunchKernelGGL((roi_prep_merged_kernel<7>), g, dim3(kBandThreads), lds, st, A, *bplan, nblocks);
    else hipLaunchKernelGGL((roi_prep_merged_kernel<14>), g, dim3(kBandThreads), lds, st, A, *bplan, nblocks);
  } else {
    if (pool == 7) hipLaunchKernelGGL((roi_fwd_prep_kernel<7>), dim3(nblocks), dim3(kBandThreads), 0, st, A);
    else hipLaunchKernelGGL((roi_fwd_prep_kernel<14>), dim3(nblocks), dim3(kBandThreads), 0, st, A);
  }
  SD_LAUNCH_CHECK();
  return SD_OK;
}

int launch_bwd_lists(const BwdFusedArgs& a, int units, hipStream_t st) {
  const size_t lds = (size_t)(a.R + 8 + 16 + a.pix_bound_words) * 4;
  const dim3 g((unsigned)units * (kListSplit + (a.pix_bound_words > 0 ? 1 : 0)));
  if (a.PP == 49) hipLaunchKernelGGL((roi_align_bwd_lists<7, 7>), g, dim3(512), lds, st, a);
  else hipLaunchKernelGGL((roi_align_bwd_lists<14, 14>), g, dim3(512), lds, st, a);
  SD_LAUNCH_CHECK();
  return SD_OK;
}

}  // namespace sd
