// ROIAlign_v2 forward/backward for gfx950 (MI355X), single level and fused FPN.
//
// Semantics follow the reference bit for bit (build with -ffp-contract=off, IEEE divide/sqrt):
//   forward   operator_cxx/contrib/roi_align_v2-inl.h:61-153 (max over the interior sample grid of
//             each bin, float argmax (x,y) stored); mixed float/double loop bounds kept (:120-125)
//   backward  operator_cxx/contrib/roi_align_v2.cu:35-84 (GPU scatter semantics)
//   assign    models/FPN/assign_layer_fpn.py:17-41
//
// MI355X design:
//  * forward: one workgroup = one RoI x G channels.  The bilinear sample grid is separable: the
//    rows/columns a RoI touches are two short index lists (<= 4*PH rows, 4*PW cols).  Lanes fill an
//    LDS tile  tile[c][row][col] = data[c][rowidx[row]][colidx[col]]  with dense, line-friendly
//    global loads (14 independent loads in flight per lane), then every lane owns one (channel,bin)
//    output, reads its 16 taps from LDS and writes out/argmax with fully contiguous stores.
//    Blocks are ordered so that each XCD works on its own channel slice (private-L2 reuse).
//  * backward: one workgroup = (image, row band, CPB channels) of ONE level.  The gradient plane
//    lives in LDS (up to 160 KB/CU on CDNA4), RoI bins are scattered into it with LDS float atomics
//    and the plane is written to HBM exactly once with coalesced 16-B stores: no zero-fill pass, no
//    global atomics.  (The reference zero-fills dX and issues 4 global atomics per output.)
//  * naive kernels (one thread per output, the reference's structure) are kept for unusual pooled
//    sizes, as the in-kernel fallback for degenerate sample loops, and as the A/B baseline.
#include "roi_align_common.h"

namespace sd {

__global__ __launch_bounds__(256) void roi_align_fwd_naive(FwdArgs a) {
  const int PP = a.PH * a.PW;
  const long count = (long)a.B * a.R * a.C * PP;
  for (long index = (long)blockIdx.x * blockDim.x + threadIdx.x; index < count;
       index += (long)gridDim.x * blockDim.x) {
    int pw = (int)(index % a.PW);
    int ph = (int)((index / a.PW) % a.PH);
    int c = (int)((index / PP) % a.C);
    int n = (int)(index / PP / a.C);
    const float* r = a.rois + (long)n * 4;
    float x1 = r[0], y1 = r[1], x2 = r[2], y2 = r[3];
    int lvl = 0;
    if (a.L.nlvl > 1) lvl = fpn_level(x1, y1, x2, y2, a.L);
    FwdOut o{0.f, -1.f, -1.f, 255};
    if (lvl >= 0) {
      int H = a.L.H[lvl], W = a.L.W[lvl];
      const float* plane = a.L.data[lvl] + ((long)(n / a.R) * a.C + c) * H * W;
      o = roi_align_fwd_elem(plane, H, W, x1, y1, x2, y2, a.L.scale[lvl], ph, pw, a.PH, a.PW);
    }
    if (a.L.nlvl > 1) o.val = o.val + 0.0f;  // add_n with the other levels' zeros
    a.out[index] = o.val;
    if (a.amax8) {
      a.amax8[((long)n * a.C + c) * amax_stride(PP) + ph * a.PW + pw] = (unsigned char)o.code;
    } else {
      a.ax[index] = o.ax;
      a.ay[index] = o.ay;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// tiled forward
// ------------------------------------------------------------------------------------------------
// One workgroup (8 waves) = NROI consecutive RoIs x a slice of the channels.
//   tables   (once per workgroup, all NROI RoIs at the same time: wave pair (2i, 2i+1) does the
//            row / column sample lists of RoI i) -> row/col offsets; per (bin,k,l) the four
//            bilinear weight products and the sample coordinates, shared by every channel
//   waves    after the tables there is NO workgroup barrier: wave w owns channels w, w+8, ... of
//            the slice and a private LDS tile.  Per (RoI, channel) it stages the RoI's taps
//            tile[row][col] = data[c][rowidx[row]][colidx[col]] with line-friendly 8-byte global
//            loads (several tile rows per wave instruction), then lanes 0..PP-1 each own one bin:
//            4 samples x 4 taps out of LDS, max + argmax, one contiguous store per output tensor.
//            The next channel's loads are issued before the current channel is computed.
template <int PH, int PW, int NROI, int NWAVE_ = 8>
struct FwdSmem {
  static constexpr int NR = 4 * PH, NC = 4 * PW, PP = PH * PW, NWAVE = NWAVE_;
  // Tile layout T[2k+l][p*PW+q][dh][dw] (tap of sample (k,l) of bin (p,q), corner (dh,dw)): the
  // four taps of one sample are one 16-byte slot and consecutive bins are consecutive slots, so
  // the compute phase is one conflict-free ds_read_b128 per sample.
  // PPP: bins padded so that the stride between the four sample planes is 16 banks (mod 32)
  static constexpr int PPP = ((PP + 3) / 8) * 8 + 4, CH = 16 * PPP;
  __attribute__((aligned(16))) float tile[NWAVE * CH];
  struct Roi {
    float4 wts[4 * PP];     // [kl][bin]: (1-a)(1-b), a(1-b), (1-a)b, ab  (kl = 2k+l); x = NaN: none
    int rowoff[NR];         // row * W, or -1 for an unused slot
    int coloff[NC];
    float hval[2 * PH], alpha[2 * PH];
    float wval[2 * PW], beta[2 * PW];
    int hcnt[PH], wcnt[PW];  // -1: empty axis bin (end <= start); else sample-loop iterations
    int binflag[PP];         // 1: the bin pools something (reference !is_empty)
    int lvl;                 // assigned level, -1 none, -2 RoI index past the end
    int n;                   // RoI index
    int fb_row, fb_col;      // a sample loop ran 3 times -> exact per-element fallback
    int any_valid;
    float box[4];
  } roi[NROI];
};

__global__ __launch_bounds__(64) void roi_coords_kernel(const float* rois, int nroi, RoiLevels L,
                                                        int PH, int PW, float* coords) {
  const int n = blockIdx.x, lane = threadIdx.x;
  const float* r = rois + (long)n * 4;
  int lvl = 0;
  if (L.nlvl > 1) lvl = fpn_level(r[0], r[1], r[2], r[3], L);
  if (lvl < 0) return;
  float* c = coords + (long)n * kCoordWords * (PH + PW);
  float* taps = c + 3 * (PH + PW);
  for (int e = lane; e < 3 * PH; e += 64) {
    const float v = sample_coord(e / 3, PH, r[1], r[3], L.scale[lvl], L.H[lvl], e % 3);
    c[e] = v;
    store_tap(taps + 2 * e, v, L.H[lvl]);
  }
  for (int e = lane; e < 3 * PW; e += 64) {
    const float v = sample_coord(e / 3, PW, r[0], r[2], L.scale[lvl], L.W[lvl], e % 3);
    c[3 * PH + e] = v;
    store_tap(taps + 2 * (3 * PH + e), v, L.W[lvl]);
  }
}

// (round 5: the 64-VGPR "lean" build, the 7x7-quadrant form of 14x14 pooling and the locality order of
// the RoIs were perf variants of this FALLBACK -- the band-resident kernel below is the product path --
// and are gone; what is left is one kernel per pooled size.)
template <int PH, int PW, int NROI, bool PK, int NWAVE_ = 8>
__device__ __forceinline__ void fwd_tiled_body(const FwdArgs& a, FwdSmem<PH, PW, NROI, NWAVE_>& s,
                                               const int bid) {
  static_assert(PH == PW, "square tiles");
  constexpr int POOL = PH;
  constexpr int PPG = POOL * POOL;                    // outputs per (RoI, channel)
  constexpr int PPSG = amax_stride(PPG);
  constexpr int D = 1;  // channels in flight per wave (deeper batches measured slower)
  using S = FwdSmem<PH, PW, NROI, NWAVE_>;
  constexpr int NR = S::NR, NC = S::NC, PP = PH * PW, PPP = S::PPP, CH = S::CH, NWAVE = S::NWAVE;
  constexpr int THREADS = NWAVE * kWave;
  static_assert(2 * NROI <= NWAVE, "one wave pair per RoI for the axis tables");
  constexpr int NPAIR = NC / 2;                 // (left,right) column pairs per tile row
  constexpr int RPW = kWave / NPAIR >= 1 ? kWave / NPAIR : 1;  // tile rows per wave instruction
  static_assert(NPAIR <= kWave, "one tile row must fit a wave");
  constexpr int ACT = RPW * NPAIR;              // active lanes in the fill
  static_assert(NR % RPW == 0, "whole fill instructions");
  constexpr int ITER = NR / RPW;                // fill instructions (8-byte loads) per channel
  constexpr int CHUNK = ITER < 8 ? ITER : 8;    // loads kept in flight per lane
  constexpr int NCHUNK = (ITER + CHUNK - 1) / CHUNK;
  constexpr int NI = (PP + kWave - 1) / kWave;  // bins per lane
  constexpr bool REGW = NI == 1;                // keep the bin's 16 weights in registers
  constexpr bool CACHE_GOFF = ITER <= 8;        // keep the fill offsets in registers

  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  // block -> (RoI group, channel slice).  Consecutive blocks are the slices of one RoI group, so
  // with nslice a multiple/divisor of 8 every XCD (block b runs on XCD b % 8) only ever touches
  // its own channel slice.
  const int nslice = a.nslice;
  const int grp = bid / nslice, slice = bid % nslice;
  const int nroi_total = a.B * a.R;
  const int nch = a.C / nslice;  // channels of this workgroup
  const int cbeg = slice * nch;

  // ---- per-RoI sample tables: wave 2i rows, wave 2i+1 columns of RoI i ----
  if (wave < 2 * NROI) {
    const int i = wave >> 1, slot = grp * NROI + i;
    typename S::Roi& t = s.roi[i];
    int lvl = -2, cnt = 0, n = 0;
    if (slot < nroi_total) {
      n = slot;
      const float* r = a.rois + (long)n * 4;
      const float x1 = r[0], y1 = r[1], x2 = r[2], y2 = r[3];
      lvl = 0;
      if (a.L.nlvl > 1) lvl = fpn_level(x1, y1, x2, y2, a.L);
      if (SD_ABLATE(a, 32) && lvl != 0) lvl = -3;  // profiling build: finest level only
      if (SD_ABLATE(a, 64) && lvl == 0) lvl = -3;  // profiling build: all but the finest level
      if (lvl >= 0) {
        const int H = a.L.H[lvl], W = a.L.W[lvl];
        const float scale = a.L.scale[lvl];
        if ((wave & 1) == 0 && lane < PH) {
          cnt = axis_samples(lane, POOL, y1, y2, scale, H, W, &t.hval[2 * lane], &t.alpha[2 * lane],
                             &t.rowoff[4 * lane]);
          t.hcnt[lane] = cnt;
        } else if ((wave & 1) == 1 && lane < PW) {
          cnt = axis_samples(lane, POOL, x1, x2, scale, W, 1, &t.wval[2 * lane], &t.beta[2 * lane],
                             &t.coloff[4 * lane]);
          t.wcnt[lane] = cnt;
        }
      }
      if ((wave & 1) == 0 && lane == 0) {
        t.box[0] = x1; t.box[1] = y1; t.box[2] = x2; t.box[3] = y2;
      }
    }
    // a 3-iteration sample loop (stride within an ulp of 0.01) does not fit the 2x2 tile layout
    const int fb = __any(cnt >= 3);
    if (lane == 0) {
      if (wave & 1) t.fb_col = fb;
      else { t.fb_row = fb; t.lvl = lvl; t.any_valid = 0; t.n = n; }
    }
  }
  __syncthreads();

  // ---- per (RoI, bin, k, l): weight products and coordinates, shared by all channels ----
  for (int e = tid; e < NROI * 4 * PP; e += THREADS) {
    const int i = e / (4 * PP), tt = e % (4 * PP);
    typename S::Roi& t = s.roi[i];
    if (t.lvl < 0 || t.fb_row || t.fb_col) continue;
    const int kl = tt / PP, bin = tt % PP, p = bin / PW, q = bin % PW, k = kl >> 1, l = kl & 1;
    const bool valid = k < t.hcnt[p] && l < t.wcnt[q];
    const float al = t.alpha[2 * p + k], be = t.beta[2 * q + l];
    float4 w;
    w.x = (1 - al) * (1 - be);
    w.y = al * (1 - be);
    w.z = (1 - al) * be;
    w.w = al * be;
    if (!valid) w.x = __int_as_float(0x7fc00000);  // NaN marks "no such sample"
    t.wts[tt] = w;
    if (kl == 0) t.binflag[bin] = (t.hcnt[p] >= 0 && t.wcnt[q] >= 0) ? 1 : 0;
    if (__any(valid) && valid) t.any_valid = 1;  // benign same-value race
  }
  __syncthreads();
  if (SD_ABLATE(a, 1)) return;
  if (PK && slice == 0) {  // the sample coordinates the packed arg-max indexes, once per RoI
    for (int e = tid; e < NROI * 3 * (PH + PW); e += THREADS) {
      const int i = e / (3 * (PH + PW)), j = e % (3 * (PH + PW));
      const typename S::Roi& t = s.roi[i];
      if (t.lvl < 0) continue;
      const bool row = j < 3 * PH;
      const int jj = row ? j : j - 3 * PH, k = jj % 3;
      const int p = jj / 3;  // bin row / column
      // recomputed, not taken from hval / wval: those hold only the samples the loop reached, and
      // the table is written in full so that its content does not depend on LDS leftovers
      const int lv = t.lvl;
      const float v = row ? sample_coord(p, POOL, t.box[1], t.box[3], a.L.scale[lv], a.L.H[lv], k)
                          : sample_coord(p, POOL, t.box[0], t.box[2], a.L.scale[lv], a.L.W[lv], k);
      float* base = a.coords + (long)t.n * kCoordWords * (POOL + POOL);
      const int jg = (row ? 0 : 3 * POOL) + p * 3 + k;
      base[jg] = v;
      store_tap(base + 3 * (POOL + POOL) + 2 * jg, v, row ? a.L.H[t.lvl] : a.L.W[t.lvl]);
    }
  }

  // ===== from here on every wave runs on its own: no workgroup barrier =====
  float* tile = s.tile + wave * CH;
  // fill: lane -> (row r0 of the RPW rows of one fill instruction, column pair jp).  A pair is the
  // (left,right) taps of one column sample; they are adjacent pixels, so one 8-byte load fetches
  // both.  When they coincide (integer coordinate / clamped border) the load starts at
  // min(left, W-2) and the register is patched (dup = 1: both .x, dup = 2: both .y).
  // Tile row rr = it*RPW + r0 = 4p + 2k + dh; LDS slot of (p,q,k,l,dh) is T[2k+l][p*PW+q][dh][dw]:
  // the lane part and the `it` part of that address are separable, so after unrolling every
  // ds_write is  lane_base + immediate.
  static_assert(RPW == 4 || RPW == 2 || RPW == 1, "fill geometry");
  const bool fill_lane = lane < ACT;
  const int jp = lane % NPAIR, r0 = lane / NPAIR;
  const int fq = jp >> 1, fl = jp & 1;
  const int lane_k = RPW == 4 ? (r0 >> 1) : 0, lane_dh = RPW == 1 ? 0 : (r0 & 1);
  const int fill_base = ((2 * lane_k + fl) * PPP + fq) * 4 + lane_dh * 2;  // floats
  auto it_part = [](int it) {  // floats; compile-time after unrolling
    const int rr = it * RPW;   // r0 = 0 part
    const int p = rr >> 2, k = (rr >> 1) & 1, dh = rr & 1;
    return ((2 * k) * PPP + p * PW) * 4 + dh * 2;
  };

  // rare RoIs first (assigned to no level, or a 3-iteration sample loop), exact and simple
#pragma unroll 1
  for (int i = 0; i < NROI; ++i) {
    const typename S::Roi& t = s.roi[i];
    const int n = t.n;
    const int lvl = t.lvl;
    if (lvl == -2) break;
    if (lvl == -3) continue;
    const long obase = ((long)n * a.C + cbeg) * PPG;
    const long abase = ((long)n * a.C + cbeg) * PPSG;
    if (lvl < 0) {  // every per-level op sees a zero box
      for (int e = tid; e < nch * PP; e += THREADS) {
        const int c = e / PP, g = e % PP;
        a.out[obase + (long)c * PPG + g] = 0.f;
        if (PK) {
          a.amax8[abase + (long)c * PPSG + g] = 255;
        } else {
          a.ax[obase + (long)c * PPG + g] = -1.f;
          a.ay[obase + (long)c * PPG + g] = -1.f;
        }
      }
    } else if (t.fb_row || t.fb_col) {
      const int H = a.L.H[lvl], W = a.L.W[lvl];
      const long plane = (long)H * W;
      const float* base = a.L.data[lvl] + ((long)(n / a.R) * a.C + cbeg) * plane;
      const float scale = a.L.scale[lvl];
      for (int e = tid; e < nch * PP; e += THREADS) {
        const int c = e / PP, bin = e % PP, g = bin;
        FwdOut o = roi_align_fwd_elem(base + (long)c * plane, H, W, t.box[0], t.box[1], t.box[2],
                                      t.box[3], scale, bin / PW, bin % PW, POOL, POOL);
        if (a.L.nlvl > 1) o.val = o.val + 0.0f;
        a.out[obase + (long)c * PPG + g] = o.val;
        if (PK) {
          a.amax8[abase + (long)c * PPSG + g] = (unsigned char)o.code;
        } else {
          a.ax[obase + (long)c * PPG + g] = o.ax;
          a.ay[obase + (long)c * PPG + g] = o.ay;
        }
      }
    }
  }

#pragma unroll 1
  for (int i = 0; i < NROI; ++i) {
    const typename S::Roi& t = s.roi[i];
    // wave-uniform values are forced into SGPRs so that every global access below is
    // "SGPR base + 32-bit lane offset" (no 64-bit vector address arithmetic in the loop)
    const int n = __builtin_amdgcn_readfirstlane(t.n);
    const int lvl = __builtin_amdgcn_readfirstlane(t.lvl);
    if (lvl == -2) break;
    if (lvl < 0 || __builtin_amdgcn_readfirstlane(t.fb_row | t.fb_col)) continue;  // (-3: ablated)
    const long obase = ((long)n * a.C + cbeg) * PPG;
    const long abase = ((long)n * a.C + cbeg) * PPSG;
    const int W = a.L.W[lvl];
    const long plane = (long)a.L.H[lvl] * W;
    const float* base = a.L.data[lvl] + ((long)(n / a.R) * a.C + cbeg) * plane;
    const long pstep = (long)NWAVE * plane * 4;  // bytes between this wave's channels

    if (!__builtin_amdgcn_readfirstlane(t.any_valid)) {  // nothing to pool anywhere in the RoI
      for (int c = wave; c < nch; c += NWAVE) {
        const long ob = obase + (long)c * PPG;
#pragma unroll
        for (int b = 0; b < NI; ++b) {
          const int bin = lane + b * kWave;
          if (bin < PP) {
            const int g = bin;
            a.out[ob + g] = 0.f;
            if (PK) {
              a.amax8[abase + (long)c * PPSG + g] = 255;
            } else {
              a.ax[ob + g] = -1.f;
              a.ay[ob + g] = -1.f;
            }
          }
        }
      }
      continue;
    }

    // ---- per-lane constants of this RoI ----
    int dup = 0;
    unsigned colbyte = 0;
    bool colok = false;
    if (fill_lane) {
      const int cl = t.coloff[2 * jp], cr = t.coloff[2 * jp + 1];
      if (cl >= 0) {
        const int co = cl == cr ? (cl < W - 1 ? cl : W - 2) : cl;
        dup = cl == cr ? (co == cl ? 1 : 2) : 0;
        colbyte = (unsigned)co * 4u;
        colok = true;
      }
    }
    const bool any_dup = __any(dup != 0);
    // byte offset of this lane's pair of fill instruction `it` within one channel plane; unused
    // slots read elements 0,1 of the plane (always in bounds, never consumed)
    auto calc_voff = [&](int it) -> unsigned {
      const int ro = fill_lane ? t.rowoff[it * RPW + r0] : -1;
      return (ro >= 0 && colok) ? (unsigned)ro * 4u + colbyte : 0u;
    };
    unsigned voff[CACHE_GOFF ? ITER : 1];
    if (CACHE_GOFF) {
#pragma unroll
      for (int it = 0; it < ITER; ++it) voff[it] = calc_voff(it);
    }
    float init[NI], cx[NI][2], cy[NI][2];
    float4 wreg[REGW ? 4 : 1];
#pragma unroll
    for (int b = 0; b < NI; ++b) {
      const int bin = lane + b * kWave;
      const int bb = bin < PP ? bin : 0, p = bb / PW, q = bb % PW;
      init[b] = t.binflag[bb] ? -FLT_MAX : 0.f;
      cx[b][0] = t.wval[2 * q]; cx[b][1] = t.wval[2 * q + 1];
      cy[b][0] = t.hval[2 * p]; cy[b][1] = t.hval[2 * p + 1];
    }
    if (REGW) {
      const int bb = lane < PP ? lane : 0;
#pragma unroll
      for (int kl = 0; kl < 4; ++kl) wreg[kl] = t.wts[kl * PP + bb];
    }

    // D channels of this wave are in flight at once (the kernel is latency bound: a tile is
    // ~3 KB of taps behind ~1 us of loaded latency, and the arithmetic per tile is ~40 VALU).
    // Loads of a batch are issued back to back, tiles are then consumed in issue order.
    float2 nxt[D][CHUNK];
    auto issue = [&](const char* pl, int d, int chunk) {
#pragma unroll
      for (int u = 0; u < CHUNK; ++u) {
        const int it = chunk * CHUNK + u;
        if (it < ITER) {
          if (SD_ABLATE(a, 2) && it >= 2) continue;  // profiling build: 2 of the ITER tap loads only
          const F2u v = *reinterpret_cast<const F2u*>(pl + (CACHE_GOFF ? voff[it] : calc_voff(it)));
          nxt[d][u] = make_float2(v.x, v.y);
        }
      }
    };
    auto commit = [&](int d, int chunk, auto dup_tag) {
      constexpr bool kDup = decltype(dup_tag)::value;
      if (fill_lane) {
#pragma unroll
        for (int u = 0; u < CHUNK; ++u) {
          const int it = chunk * CHUNK + u;
          if (it < ITER) {
            float2 v = nxt[d][u];
            if (kDup) {
              if (dup == 1) v.y = v.x;
              if (dup == 2) v.x = v.y;
            }
            *reinterpret_cast<float2*>(tile + fill_base + it_part(it)) = v;
          }
        }
      }
    };

    // wave w handles channels w, w+NWAVE, ... of the slice.  The loop is instantiated twice so
    // that the (rare) coincident-column patch costs nothing on the common path.
    auto channel_loop = [&](auto dup_tag) {
      const char* pl = reinterpret_cast<const char*>(base + (long)wave * plane);
      float* po = a.out + obase + (long)wave * PPG;
      float* px = PK ? nullptr : a.ax + obase + (long)wave * PPG;
      float* py = PK ? nullptr : a.ay + obase + (long)wave * PPG;
      unsigned char* pk = PK ? a.amax8 + abase + (long)wave * PPSG : nullptr;
      for (int c0 = wave; c0 < nch; c0 += D * NWAVE) {
#pragma unroll
        for (int d = 0; d < D; ++d)
          if (c0 + d * NWAVE < nch) issue(pl + d * pstep, d, 0);
#pragma unroll
        for (int d = 0; d < D; ++d) {
          if (c0 + d * NWAVE < nch) {
            commit(d, 0, dup_tag);
#pragma unroll
            for (int ch = 1; ch < NCHUNK; ++ch) {
              issue(pl + d * pstep, d, ch);
              commit(d, ch, dup_tag);
            }
            wave_lds_sync();  // the tile was written by the fill lanes, read by the bin lanes
#pragma unroll
            for (int b = 0; b < NI; ++b) {
              const int bin = lane + b * kWave;
              if (bin < PP) {
                float maxval = init[b], bx = -1.f, by = -1.f;
                int bk = -1;
                const float4* tp = reinterpret_cast<const float4*>(tile) + bin;
#pragma unroll
                for (int kl = 0; kl < 4; ++kl) {
                  // an absent sample has w.x = NaN: its value is NaN and never wins the comparison
                  const float4 w = REGW ? wreg[kl] : t.wts[kl * PP + bin];
                  const float4 v = tp[kl * PPP];  // (TL, TR, BL, BR)
                  const float value = w.x * v.x + w.y * v.z + w.z * v.y + w.w * v.w;
                  if (value > maxval) {
                    maxval = value;
                    if (PK) {
                      bk = (kl >> 1) * 3 + (kl & 1);
                    } else {
                      bx = cx[b][kl & 1];
                      by = cy[b][kl >> 1];
                    }
                  }
                }
                if (a.L.nlvl > 1) maxval = maxval + 0.0f;
                const int g = bin;
                po[g + d * NWAVE * PPG] = maxval;
                if (PK) {
                  if (!(SD_ABLATE(a, 4))) pk[g + d * NWAVE * PPSG] = (unsigned char)(bk < 0 ? 255 : bk);
                } else {
                  px[g + d * NWAVE * PPG] = bx;
                  py[g + d * NWAVE * PPG] = by;
                }
              }
            }
            wave_lds_sync();  // ... and is refilled for the next channel
          }
        }
        pl += D * pstep;
        po += D * NWAVE * PPG;
        if (PK) {
          pk += D * NWAVE * PPSG;
        } else {
          px += D * NWAVE * PPG;
          py += D * NWAVE * PPG;
        }
      }
    };
    if (any_dup) channel_loop(std::true_type{});
    else channel_loop(std::false_type{});
  }
}

template <int PH, int PW, int NROI, bool PK>
__global__ __launch_bounds__(512) void roi_align_fwd_tiled(FwdArgs a) {
  __shared__ FwdSmem<PH, PW, NROI> s;
  fwd_tiled_body<PH, PW, NROI, PK>(a, s, blockIdx.x);
}

// ------------------------------------------------------------------------------------------------
// Band-resident forward (round 3, the default).  The dual of the backward: the feature planes are
// streamed through LDS by dense 16-byte loads (every byte read from HBM once, plus a halo), and
// the RoI bins read their taps out of LDS -- no per-RoI global gathers at all.
//   unit        (level, image, row band): the band's rows [r0, r0 + owned) plus kBandHalo rows
//               below it; a level whose plane fits one buffer is a single band holding G planes
//   item        (RoI, bin row p) -- assigned to the band that holds the first tap row of the bin
//               row; a RoI is eligible when every bin row's taps span <= halo + 1 rows (all RoIs
//               on their FPN level are; the others -- and 3-iteration sample loops, and RoIs of no
//               level -- go to a few exact per-element workgroups at the end of the launch)
//   pre-pass    one launch: per unit the item list (RoI | p << 16), per RoI and axis bin a
//               16-byte entry {neighbour offsets, validity flags, interpolation fractions}
//   workgroup   (unit, chunk of `steps` fills): 16 waves, one per CU (2 x 66 KB buffers).  Every
//               wave keeps the addresses / fractions of its passes (9 items x 7 bins = 63 lanes
//               each) in registers across the whole channel loop, so the per-channel work is
//               eight ds_read2_b32 + the reference's arithmetic + two stores per pass; the next
//               channel's band is already on its way into the other buffer (global_load_lds).
// Same float expressions in the same order as roi_align_fwd_elem: bit-equal results.
// ------------------------------------------------------------------------------------------------
// dense copy of `len` floats at gsrc into LDS at buf (+ shift floats: the 16-byte misalignment of
// gsrc), by global_load_lds_dwordx4 (LDS destination = wave-uniform base + lane * 16).  Returns the
// shift.  The (at most two) partial 16-byte words at the ends are fetched as single floats.
__device__ __forceinline__ int band_fill(const float* gsrc, int len, float* buf, int wave, int lane) {
  const int shift = (int)(((uintptr_t)gsrc >> 2) & 3);
  const float* a0 = gsrc - shift;                        // 16-byte aligned
  const int n4 = (shift + len + 3) >> 2;                 // 16-byte words that hold the band
  const int first_full = shift ? 1 : 0;
  const int last_full = ((shift + len) >> 2);            // exclusive
  const float4* s4 = reinterpret_cast<const float4*>(a0);
  float4* d4 = reinterpret_cast<float4*>(buf);
  for (int w4 = wave * kWave; w4 < last_full; w4 += kBandWaves * kWave) {
    const int i = w4 + lane;
    if (i >= first_full && i < last_full) __builtin_amdgcn_global_load_lds(s4 + i, d4 + w4, 16, 0, 0);
  }
  if (wave == 0) {
    if (shift && lane < 4 && lane >= shift && lane < shift + len)
      __builtin_amdgcn_global_load_lds(a0 + lane, buf, 4, 0, 0);
    if (last_full < n4 && last_full >= first_full && (last_full > 0 || !shift)) {
      const int j = last_full * 4 + lane;
      if (lane < 4 && j < shift + len) __builtin_amdgcn_global_load_lds(a0 + j, buf + last_full * 4, 4, 0, 0);
    }
  }
  return shift;
}

// HALF: the feature maps and the output are fp16 (the arithmetic stays fp32: the taps are converted
// on their way into LDS, the maximum is rounded to nearest even on the way out) -- what an fp16 graph
// gets from X.to_fp32 -> ROIAlign -> X.to_fp16 (models/FPN/builder.py:581-586, 607-608) without the
// two cast passes, and with half the feature traffic.
template <int POOL, bool PK, bool HALF = false>
__global__ __launch_bounds__(kBandThreads) void roi_align_fwd_band(BandArgs A) {
  using TIn = typename std::conditional<HALF, __half, float>::type;
  constexpr int AL = HALF ? 8 : 4;   // elements per 16 bytes of the input
  const FwdArgs& a = A.f;
  const BandPlan& P = A.p;
  constexpr int QL = POOL, IPP = kWave / QL;             // lanes per item, items per pass
  // passes per wave and round (the float arg-max form carries four sample coordinates more per
  // pass, the fp16 form twelve staging registers)
  constexpr int NP = (PK && !HALF) ? kBandNP : kBandNP - 1, CAP = NP * kBandWaves * IPP;
  constexpr int PPG = POOL * POOL, PPSG = amax_stride(PPG);
  extern __shared__ __attribute__((aligned(16))) float band_smem[];
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);

#ifdef SD_PROFILING
  const long long t_entry = __builtin_readcyclecounter();
#endif
  // ---- persistent workgroups, one per CU.  Work = (virtual unit, channel): a virtual unit is a
  // unit's items cut into rounds of CAP (what one workgroup keeps in registers).  A workgroup
  // starts on the virtual unit its share of the estimated cost falls in, takes the unit's
  // channels G at a time from a per-unit counter (so the workgroups of a unit finish together
  // whatever the estimate was worth), and when the unit runs dry moves to the unit with the most
  // work left ----
  __shared__ int v_unit[kBandMaxUnits], v_first[kBandMaxUnits], v_items[kBandMaxUnits];
  __shared__ int v_cost[kBandMaxUnits], v_start[kBandMaxUnits + 1];
  __shared__ int2 s_grab[2];  // reservations {first channel, channels}
  __shared__ int s_pick;
  const int wg = (int)blockIdx.x - kBandFallbackWGs, nwg = (int)gridDim.x - kBandFallbackWGs;
  const int rsub = (a.R + kBandSub - 1) / kBandSub;
  auto level_of = [&](int u) {
    int l = 0;
    for (int k = 0; k < a.L.nlvl; ++k)
      if (a.L.stride[k] >= 0 && u >= P.unit_base[k]) l = k;
    return l;
  };
  // The table of virtual units and their cost prefix: one wave does it (lane = unit, 64 at a time,
  // wave scans: no workgroup barriers on the start-up path), the others wait at one barrier.
  __shared__ int s_nvu;
  auto wave_incl_scan = [&](int v) {
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const int t = __shfl_up(v, o);
      if (lane >= o) v += t;
    }
    return v;
  };
  // (round 4: wave b builds the entries of units 64 b .. 64 b + 63 -- one global round trip and two wave
  // scans per WAVE instead of per 64 units of one wave's serial walk, ~10 % of a workgroup's life at the
  // baseline's 257 units; the cross-wave offsets go through LDS)
  static_assert(kBandMaxUnits % kWave == 0 && kBandMaxUnits / kWave <= kBandWaves, "unit table: one wave per 64 units");
  constexpr int NB = kBandMaxUnits / kWave;
  __shared__ int s_tot[NB], s_ctot[NB];
  {
    // phase A: rounds per unit, scanned inside the wave
    const int u = wave * kWave + lane;
    int cnt = 0, rounds = 0, lv = 0, incl = 0;
    if (wave < NB) {
      if (u < P.nunits) {
        lv = level_of(u);
#pragma unroll
        for (int j = 0; j < kBandSub; ++j) cnt += P.seg[u * kBandSub + j].y;
        rounds = (cnt + CAP - 1) / CAP;
      }
      incl = wave_incl_scan(rounds);
      if (lane == kWave - 1) s_tot[wave] = incl;
    }
    __syncthreads();
    // phase B: the virtual units of this wave's units, behind those of the waves before it
    int total = 0;
#pragma unroll
    for (int b = 0; b < NB; ++b) total += s_tot[b];
    const int n = total < kBandMaxUnits ? total : kBandMaxUnits;  // (the launcher keeps a level's rounds few)
    if (wave < NB) {
      int vb = incl - rounds;
#pragma unroll
      for (int b = 0; b < NB; ++b) vb += b < wave ? s_tot[b] : 0;
      for (int r = 0; r < rounds && vb + r < kBandMaxUnits; ++r) {
        const int it = cnt - r * CAP < CAP ? cnt - r * CAP : CAP;
        v_unit[vb + r] = u;
        v_first[vb + r] = r * CAP;
        v_items[vb + r] = it;
        // measured (profiles/r03_fwd_cost_model.txt): a fill takes ~ F0 + G * (F1 + k * items) ticks,
        // linear in the items (the waves of a SIMD share the LDS and VALU rate); per channel, in units of k:
        // (units whose fills hold several planes hand work out in coarser pieces: they get a larger
        // share of the workgroups, finish early and their workgroups then join the fine-grained units)
        const int cst = kBandFillCost / P.g[lv] + kBandPlaneCost + it;
        v_cost[vb + r] = P.g[lv] > 1 ? cst + cst * P.gbias / 100 : cst;
      }
    }
    __syncthreads();
    // phase C: exclusive prefix of the virtual units' cost, wave w for virtual units 64 w .. 64 w + 63
    const int v = wave * kWave + lane;
    int mine = 0, cincl = 0;
    if (wave < NB) {
      mine = v < n ? kBandSetupCost + v_cost[v] * a.C : 0;
      cincl = wave_incl_scan(mine);
      if (lane == kWave - 1) s_ctot[wave] = cincl;
    }
    __syncthreads();
    if (wave < NB) {
      int run = cincl - mine;
#pragma unroll
      for (int b = 0; b < NB; ++b) run += b < wave ? s_ctot[b] : 0;
      if (v < n) v_start[v] = run;
      if (v == n - 1 || (n == 0 && v == 0)) v_start[n] = n == 0 ? 0 : run + mine;
    }
    if (tid == 0) s_nvu = n;
  }
  __syncthreads();
  const int nvu = s_nvu;
  int vu = 0;
  {
    // the last virtual unit whose start is <= this workgroup's share of the cost (v_start ascends)
    const long pos = (long)v_start[nvu] * (2 * wg + 1) / (2 * nwg);
    int below = 0;
    for (int k = lane; k < nvu; k += kWave) below += v_start[k] <= pos ? 1 : 0;
    below = wave_sum_i32(below);
    vu = below > 0 ? below - 1 : 0;
  }
  float* buf0 = band_smem;
  float* buf1 = band_smem + kBandBufFloats;
#ifdef SD_PROFILING
  const long long t_begin = __builtin_readcyclecounter();
  long long t_setup = 0, t_wait = 0, t_comp = 0, t_mark = t_begin;
  int dbg_count = 0, dbg_units = 0, dbg_fills = 0;
#endif
  // takes the next (up to) G channels of virtual unit v: first channel, or >= C when it has run dry
  auto grab = [&](int v, int G) {
    int k = 0;
    if (tid == 0) k = atomicAdd(&P.chan_ctr[v], G);
    return k;  // (valid in thread 0 only)
  };

  // A workgroup keeps visiting virtual units until every unit's channel counter has been taken past
  // C: a unit somebody has grabbed from is finished by its visitors (they loop until the counter runs
  // dry), so the launch is complete exactly when no unit is left with an untouched counter.
  for (int visit = 0; nvu > 0; ++visit) {
  if (visit) {
    // the unit ran dry: move to the virtual unit with the most estimated work left (if any);
    // one wave looks (fresh counter values), one barrier
    if (wave == 0) {
      int best = -1, bestval = 0;
      for (int v0 = 0; v0 < nvu; v0 += kWave) {
        const int v = v0 + lane;
        int left = 0;
        if (v < nvu) {
          const int done = __hip_atomic_load(&P.chan_ctr[v], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          // (joining costs a set-up: only worth it for a few fills -- but a unit nobody has started
          // yet must be taken by somebody, however small it is)
          if (done < a.C && (done == 0 || a.C - done >= 3 * P.g[level_of(v_unit[v])])) {
            left = (a.C - done) * v_cost[v];
            left = left < 1 ? 1 : left;
          }
        }
        int m = left;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
          const int t = __shfl_xor(m, o);
          m = t > m ? t : m;
        }
        if (m > bestval) {
          bestval = m;
          best = v0 + __builtin_ctzll(__ballot(left == m));
        }
      }
      if (lane == 0) s_pick = best;
    }
    __syncthreads();
    vu = s_pick;
    if (vu < 0) break;
  }
  const int unit = v_unit[vu];
  const int lvl = level_of(unit);
  const int G = P.g[lvl], nb = P.nbands[lvl];
  // channels are reserved GR at a time (a run of consecutive channels: the (RoI, channel) rows of the
  // outputs are 196 bytes, neighbours share cache lines, and a run written by one CU merges in its L2)
  // In the last P.tail per cent of a unit reservations shrink to single fills, so that what a
  // workgroup still holds when the counter runs dry is small.
  const int GR = ((P.grab + G - 1) / G) * G;
  int glast = 0;  // (thread 0) first channel of the last reservation it got
  auto next_size = [&]() { return glast >= a.C - a.C * P.tail / 100 ? (P.tail_planes < G ? P.tail_planes : G) : GR; };
  {
    const int k = grab(vu, GR);
    if (tid == 0) s_grab[0] = make_int2(k, GR);
    glast = k;
    __syncthreads();
  }
  int kcur = s_grab[0].x;
  if (kcur >= a.C) continue;  // (uniform) dry already
  int ck = kcur + G, cend = kcur + GR < a.C ? kcur + GR : a.C;   // rest of the current reservation
  int gcur = cend - kcur < G ? cend - kcur : G, gnext = 0;       // planes of the current / next fill
  int slot = 1;                                                  // where the next reservation is parked
  const int ul = unit - P.unit_base[lvl];
  const int img = ul / nb, band = ul % nb;
  const int H = a.L.H[lvl], W = a.L.W[lvl], HW = H * W;
  const int r0 = band * P.owned[lvl];
  const int nrows = H - r0 < P.rows[lvl] ? H - r0 : P.rows[lvl];
  const int blen = nrows * W;                            // floats of one plane's band
  // LDS floats between the G planes of a fill (fp16: whole 8-element words, whatever the misalignment)
  const int bstride = HALF ? ((blen + 14) >> 3) * 8 : (blen + 4 + 3) & ~3;
  // the unit's list = kBandSub segments, one per quarter of the image's RoIs
  const unsigned* items = P.items + ((long)img * SD_MAX_FPN_LEVELS + lvl) * kBandSub * rsub * POOL;
  int sgs[kBandSub], sgc[kBandSub], count = 0;
#pragma unroll
  for (int j = 0; j < kBandSub; ++j) {
    const int2 sg = P.seg[unit * kBandSub + j];
    sgs[j] = j * rsub * POOL + sg.x - count;   // items[sgs[j] + t] for list positions t of segment j
    sgc[j] = count + sg.y;                     // (exclusive end of segment j in list positions)
    count += sg.y;
  }
  const int round0 = v_first[vu], nitems = v_items[vu];
#ifdef SD_PROFILING
  dbg_count += nitems;
  ++dbg_units;
  dbg_fills = 0;
  t_mark = __builtin_readcyclecounter();
  if (a.dbg && lane == 0 && wave == 0 && dbg_units <= 3) {  // per visit: level, fills, items, start tick
    long long* d = a.dbg + ((long)gridDim.x * kBandWaves + (long)blockIdx.x * 4 + (dbg_units - 1)) * 8;
    d[0] = lvl; d[2] = nitems; d[3] = t_mark; d[4] = 1;
  }
#endif
  const TIn* gbase = reinterpret_cast<const TIn*>(a.L.data[lvl]) + (long)img * a.C * HW + (long)r0 * W;  // channel 0 of the band
  {
    // one fill = the band rows of G consecutive planes; plane g lands at g * bstride (+ its shift)
    // fp32: straight into LDS (global_load_lds).  fp16: 16-byte words into registers when the fill
    // is issued, converted and stored to LDS after the step's arithmetic (fill_commit).
    constexpr int NST = HALF ? 3 : 1;   // staged 16-byte words per thread (<= kBandBufFloats / 8 / 1024 + 1)
    uint4 st[NST];
    const int n8u = (blen + 7 + 7) >> 3;  // words per plane at most (any misalignment)
    auto fill = [&](const TIn* src, float* dst, int gcount) {
      if constexpr (!HALF) {
        int sh0 = 0;
        for (int g = 0; g < gcount; ++g) {
          const int sh = band_fill(src + (long)g * HW, blen, dst + g * bstride, wave, lane);
          if (g == 0) sh0 = sh;
        }
        return sh0;
      } else {
        const int sh0 = (int)(((uintptr_t)src >> 1) & 7);
#pragma unroll
        for (int k = 0; k < NST; ++k) {
          const int f = tid + k * kBandThreads, g = f / n8u, i = f - g * n8u;
          st[k] = make_uint4(0, 0, 0, 0);
          if (g < gcount) {
            const __half* sp = src + (long)g * HW;
            const int sh = (int)(((uintptr_t)sp >> 1) & 7);
            const int e0 = 8 * i - sh;                 // band element of the word's first half
            if (e0 >= 0 && e0 + 8 <= blen) {
              st[k] = *reinterpret_cast<const uint4*>(sp + e0);
            } else if (e0 + 8 > 0 && e0 < blen) {      // a word that sticks out of the band: by halves
              unsigned short h[8];
#pragma unroll
              for (int j = 0; j < 8; ++j)
                h[j] = (e0 + j >= 0 && e0 + j < blen) ? reinterpret_cast<const unsigned short*>(sp)[e0 + j] : 0;
              st[k] = make_uint4(h[0] | (unsigned)h[1] << 16, h[2] | (unsigned)h[3] << 16,
                                 h[4] | (unsigned)h[5] << 16, h[6] | (unsigned)h[7] << 16);
            }
          }
        }
        return sh0;
      }
    };
    auto fill_commit = [&](float* dst, int gcount) {
      if constexpr (HALF) {
#pragma unroll
        for (int k = 0; k < NST; ++k) {
          const int f = tid + k * kBandThreads, g = f / n8u, i = f - g * n8u;
          if (g < gcount) {
            const __half2* hp = reinterpret_cast<const __half2*>(&st[k]);
            const float2 p0 = __half22float2(hp[0]), p1 = __half22float2(hp[1]);
            const float2 p2 = __half22float2(hp[2]), p3 = __half22float2(hp[3]);
            float4* d = reinterpret_cast<float4*>(dst + g * bstride + 8 * i);
            d[0] = make_float4(p0.x, p0.y, p1.x, p1.y);
            d[1] = make_float4(p2.x, p2.y, p3.x, p3.y);
          }
        }
      }
    };
    // ---- per-pass state, in registers across the channel loop.  The table loads go out before
    // the first fill (loads return in order: behind the fill they would wait for all of it), the
    // arithmetic on them runs while the fill lands ----
    int A0[NP], A1[NP], A2[NP], A3[NP], A4[NP], A5[NP], A6[NP], A7[NP];
    float al0[NP], al1[NP], be0[NP], be1[NP], cx0[NP], cx1[NP], cy0[NP], cy1[NP];
    int ooff[NP];       // element index of the bin in out (first channel of the chunk)
    unsigned aoff[NP];  // byte index of its arg-max code
    int flags[NP];      // bit 0 valid lane, 1 dup0, 2 dup1, 3 empty
    bool anyd[NP];
    unsigned words[NP];
    uint4 res[NP], ces[NP];
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int it = (wave + kBandWaves * i) * IPP + lane / QL;
      const bool valid = lane < IPP * QL && it < nitems;
      words[i] = 0;
      if (valid) {
        const int t = round0 + it;
        int o = sgs[kBandSub - 1];
#pragma unroll
        for (int j = kBandSub - 2; j >= 0; --j) o = t < sgc[j] ? sgs[j] : o;
        words[i] = items[o + t];
      }
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int it = (wave + kBandWaves * i) * IPP + lane / QL;
      const bool valid = lane < IPP * QL && it < nitems;
      const int n = words[i] & 0xffff, pp = words[i] >> 16, q = lane % QL;
      res[i] = make_uint4(0, 0, 0x7fc00000u, 0x7fc00000u);
      ces[i] = res[i];
      if (valid) {
        res[i] = P.rowent[((long)img * a.R + n) * POOL + pp];
        ces[i] = P.colent[((long)img * a.R + n) * POOL + q];
      }
    }
    int shift_next = fill(gbase + (long)kcur * HW, buf0, gcur);
    fill_commit(buf0, gcur);   // (fp16: the first fill is not hidden)
    if (tid == 0) {  // the reservation after this one (read past the next barrier)
      const int sz = next_size();
      glast = grab(vu, sz);
      s_grab[1] = make_int2(glast, sz);
    }
#pragma unroll
    for (int i = 0; i < NP; ++i) {
      const int it = (wave + kBandWaves * i) * IPP + lane / QL;
      const bool valid = lane < IPP * QL && it < nitems;
      const unsigned word = words[i];
      const int n = word & 0xffff, pp = word >> 16, q = lane % QL;
      const uint4 re = res[i], ce = ces[i];
      if (!PK) {
        float2 rv = make_float2(0.f, 0.f), cv = rv;
        if (valid) {
          rv = P.rowval[((long)img * a.R + n) * POOL + pp];
          cv = P.colval[((long)img * a.R + n) * POOL + q];
        }
        cy0[i] = rv.x; cy1[i] = rv.y; cx0[i] = cv.x; cx1[i] = cv.y;
      }
      const int r0w = r0 * W;
      int lo0 = (int)(re.x & 0xfffff) - r0w, lo1 = (int)(re.y & 0xfffff) - r0w;
      lo0 = lo0 < 0 ? 0 : lo0;   // (an absent sample has offset 0: keep its unused address in range)
      lo1 = lo1 < 0 ? 0 : lo1;
      const int hi0 = lo0 + ((re.x >> 20) & 1 ? W : 0), hi1 = lo1 + ((re.y >> 20) & 1 ? W : 0);
      const int left0 = ce.x & 0xfff, left1 = (ce.x >> 13) & 0xfff;
      A0[i] = (lo0 + left0) * 4; A1[i] = (lo0 + left1) * 4; A2[i] = (hi0 + left0) * 4; A3[i] = (hi0 + left1) * 4;
      A4[i] = (lo1 + left0) * 4; A5[i] = (lo1 + left1) * 4; A6[i] = (hi1 + left0) * 4; A7[i] = (hi1 + left1) * 4;
      al0[i] = __uint_as_float(re.z); al1[i] = __uint_as_float(re.w);
      be0[i] = __uint_as_float(ce.z); be1[i] = __uint_as_float(ce.w);
      const int d0 = (ce.x >> 12) & 1, d1 = (ce.x >> 25) & 1;
      const int empty = (int)(re.x >> 31) | (int)((ce.x >> 26) & 1);
      flags[i] = (valid ? 1 : 0) | d0 << 1 | d1 << 2 | empty << 3;
      anyd[i] = __ballot(valid && (d0 | d1)) != 0;
      ooff[i] = (int)((((long)img * a.R + n) * a.C) * PPG + pp * POOL + q);   // (channel 0)
      aoff[i] = (unsigned)((((long)img * a.R + n) * a.C) * PPSG + pp * POOL + q);
    }

    for (int s = 0;; ++s) {
      // this wave's share of fill s has landed (hipcc does not count global_load_lds against the
      // barrier by itself); after the barrier everyone's has, and everyone is done with the other buffer
#ifdef SD_PROFILING
      {
        const long long now = __builtin_readcyclecounter();
        if (s == 0) t_setup += now - t_mark; else t_comp += now - t_mark;
        t_mark = now;
      }
#endif
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
#ifdef SD_PROFILING
      {
        const long long now = __builtin_readcyclecounter();
        t_wait += now - t_mark;
        t_mark = now;
      }
#endif
      const int shift = shift_next;
      const int gcount = gcur;
      // next fill: the rest of this reservation, else the parked one (and a new one is requested;
      // the counter's answer stays in a register while the step computes)
      int knext = a.C, grabbed = 0;
      bool regrab = false;
      if (ck < cend) {
        knext = ck;
        ck += G;
      } else {
        const int2 b = s_grab[slot];
        if (b.x < a.C) {
          knext = b.x;
          ck = b.x + G;
          cend = b.x + b.y < a.C ? b.x + b.y : a.C;
          slot ^= 1;
          regrab = true;
        }
      }
      if (knext < a.C) {
        gnext = cend - knext < G ? cend - knext : G;   // (cend: end of the reservation knext lies in)
        shift_next = fill(gbase + (long)knext * HW, (s & 1) ? buf0 : buf1, gnext);
      }
      int gsz = 0;
      if (regrab) {
        gsz = next_size();
        grabbed = grab(vu, gsz);
      }
      const char* base = reinterpret_cast<const char*>((s & 1) ? buf1 : buf0);
#pragma unroll
      for (int i = 0; i < NP; ++i) {
        if ((wave + kBandWaves * i) * IPP >= nitems) break;   // (wave-uniform)
        const float init = (flags[i] & 8) ? 0.f : -FLT_MAX;
        // weight products, the reference's expressions (roi_align_v2-inl.h:131-134)
        // (recomputed per step from opaque copies of the fractions: hoisted out of the channel loop
        // the 16 products of every pass would occupy 16 * NP registers)
        float fa0 = al0[i], fa1 = al1[i], fb0 = be0[i], fb1 = be1[i];
        asm volatile("" : "+v"(fa0), "+v"(fa1), "+v"(fb0), "+v"(fb1));
        // products paired the way the taps arrive: ds_read2_b32 delivers (left, right) of one row,
        // so {(1-a)(1-b), (1-a)b} multiplies the low row's pair and {a(1-b), ab} the high row's in
        // one v_pk_mul_f32 each, no register shuffling
        const v2f b0 = {1 - fb0, fb0}, b1 = {1 - fb1, fb1};
        const v2f wl00 = (1 - fa0) * b0, wh00 = fa0 * b0, wl01 = (1 - fa0) * b1, wh01 = fa0 * b1;
        const v2f wl10 = (1 - fa1) * b0, wh10 = fa1 * b0, wl11 = (1 - fa1) * b1, wh11 = fa1 * b1;
        int oo = ooff[i] + kcur * PPG;
        unsigned ao = aoff[i] + (unsigned)(kcur * PPSG);
        for (int g = 0; g < gcount; ++g) {
          // plane g of the fill; its 16-byte misalignment follows from plane 0's (HW floats apart)
          const char* pl = base + ((long)g * bstride + ((shift + g * (HW & (AL - 1))) & (AL - 1))) * 4;
          auto rd = [&](int off) {
            const F2u t = *reinterpret_cast<const F2u*>(pl + off);
            return v2f{t.x, t.y};
          };
          v2f t000 = rd(A0[i]), t001 = rd(A1[i]), t010 = rd(A2[i]), t011 = rd(A3[i]);
          v2f t100 = rd(A4[i]), t101 = rd(A5[i]), t110 = rd(A6[i]), t111 = rd(A7[i]);
          if (anyd[i]) {  // coincident (left, right) columns: both taps are the left pixel
            if (flags[i] & 2) { t000.y = t000.x; t010.y = t010.x; t100.y = t100.x; t110.y = t110.x; }
            if (flags[i] & 4) { t001.y = t001.x; t011.y = t011.x; t101.y = t101.x; t111.y = t111.x; }
          }
          float maxval = init, bx_ = -1.f, by_ = -1.f;
          int bk = 255;
          // value = w1*TL + w2*BL + w3*TR + w4*BR, summed left to right (roi_align_v2-inl.h:135-138)
          auto val4 = [](v2f wl, v2f wh, v2f lo, v2f hi) {
            const v2f ml = wl * lo, mh = wh * hi;
            return ((ml.x + mh.x) + ml.y) + mh.y;
          };
          float value;
          value = val4(wl00, wh00, t000, t010);
          if (value > maxval) { maxval = value; bk = 0; if (!PK) { bx_ = cx0[i]; by_ = cy0[i]; } }
          value = val4(wl01, wh01, t001, t011);
          if (value > maxval) { maxval = value; bk = 1; if (!PK) { bx_ = cx1[i]; by_ = cy0[i]; } }
          value = val4(wl10, wh10, t100, t110);
          if (value > maxval) { maxval = value; bk = 3; if (!PK) { bx_ = cx0[i]; by_ = cy1[i]; } }
          value = val4(wl11, wh11, t101, t111);
          if (value > maxval) { maxval = value; bk = 4; if (!PK) { bx_ = cx1[i]; by_ = cy1[i]; } }
          if (a.L.nlvl > 1) maxval = maxval + 0.0f;
          if (flags[i] & 1) {
            // (profiling build, roi_align_fwd_ablate = 128 skips the value stores: -11 us, which is what
            // the same 51 MB of 28-byte rows cost alone, tools/store_bench.hip = 4.4 TB/s)
            if (!(SD_ABLATE(a, 128))) {
              if constexpr (HALF) reinterpret_cast<__half*>(a.out)[oo] = __float2half(maxval);
              else a.out[oo] = maxval;
            }
            if (SD_ABLATE(a, 256)) {   // (profiling build: no arg-max stores)
            } else if (PK) {
              a.amax8[ao] = (unsigned char)bk;
            } else {
              a.ax[oo] = bx_;
              a.ay[oo] = by_;
            }
          }
          oo += PPG;
          ao += PPSG;
        }
      }
#ifdef SD_PROFILING
      ++dbg_fills;
#endif
      if (knext < a.C) fill_commit((s & 1) ? buf0 : buf1, gnext);
      if (tid == 0 && regrab) {
        s_grab[slot] = make_int2(grabbed, gsz);
        glast = grabbed;
      }
      kcur = knext;
      gcur = gnext;
      if (kcur >= a.C) break;  // (uniform) the unit has no fill left for this workgroup
    }
#ifdef SD_PROFILING
    {
      const long long now = __builtin_readcyclecounter();
      t_comp += now - t_mark;
      t_mark = now;
    }
#endif
    __syncthreads();  // the next visit refills buf0 and reuses s_grab
  }
#ifdef SD_PROFILING
  if (a.dbg && lane == 0 && wave == 0 && dbg_units <= 3) {
    long long* d = a.dbg + ((long)gridDim.x * kBandWaves + (long)blockIdx.x * 4 + (dbg_units - 1)) * 8;
    d[1] = dbg_fills; d[5] = __builtin_readcyclecounter();
  }
#endif
  }  // visits
  {
    // ---- exact per-element path for the few RoIs the bands do not take (and the constant output
    // of the RoIs that pool nothing), after the band work: workgroup = (RoI slot, channel slice).
    // (As blocks of their own in front of the launch they delayed every band workgroup's start.) ----
    const int nroi = a.B * a.R;
    // Round 6: the RoIs that pool nothing (flag 2) of the float arg-max form first, as whole (RoI, all channels) blocks of
    // constants -- C * 49 contiguous floats per output -- with 16-byte stores, one RoI per workgroup and trip.  In the
    // un-fused FPN graph (models/FPN/builder.py:588-605) three quarters of a level op's RoIs are such rows (zero boxes of
    // the other levels): 115 MB of a 154 MB output; walked below as (RoI, channel slice) tasks of element stores, each behind
    // its own dependent load of the box, they were the tail of the launch.
    bool fast_void = false;
    if constexpr (!PK && !HALF) {
      fast_void = (((uintptr_t)a.out | (uintptr_t)a.ax | (uintptr_t)a.ay) & 15) == 0 && (a.C * PPG) % 4 == 0;
      if (fast_void) {
        const int n4 = a.C * PPG / 4;
        const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f), neg = make_float4(-1.f, -1.f, -1.f, -1.f);
        for (int n = wg; n < nroi; n += nwg) {
          if (P.fbflag[n] != 2) continue;   // (uniform)
          float4* o4 = reinterpret_cast<float4*>(a.out + (long)n * a.C * PPG);
          float4* x4 = reinterpret_cast<float4*>(a.ax + (long)n * a.C * PPG);
          float4* y4 = reinterpret_cast<float4*>(a.ay + (long)n * a.C * PPG);
          for (int i = tid; i < n4; i += kBandThreads) {
            o4[i] = zero;
            x4[i] = neg;
            y4[i] = neg;
          }
        }
      }
    }
    const int nsl = nwg >= a.fbslice ? a.fbslice : 1, csl = a.C / nsl, slice = wg % nsl;
    const int nslots = nwg / nsl;
    // the flags of this workgroup's RoIs are fetched 64 at a time by every wave (one load each,
    // not a chain of dependent loads), then only the flagged ones are visited
    for (int n0 = wg / nsl; n0 < nroi && wg / nsl < nslots; n0 += nslots * kWave) {
      const int nl = n0 + lane * nslots;
      int myflag = nl < nroi ? P.fbflag[nl] : 0;
      if (fast_void && myflag == 2) myflag = 0;   // written above
      unsigned long long todo = __ballot(myflag != 0);
      while (todo) {
      const int src = __builtin_ctzll(todo);
      todo &= todo - 1;
      const int n = n0 + src * nslots;
      const int flag = __builtin_amdgcn_readlane(myflag, src);
      const float4 bx = *reinterpret_cast<const float4*>(a.rois + (long)n * 4);
      const int lvl = a.L.nlvl > 1 ? fpn_level(bx.x, bx.y, bx.z, bx.w, a.L) : 0;
      for (int e = tid; e < csl * PPG; e += kBandThreads) {
        const int c = slice * csl + e / PPG, g = e % PPG;
        FwdOut o{0.f, -1.f, -1.f, 255};
        if (lvl >= 0 && flag == 1) {
          const int H = a.L.H[lvl], W = a.L.W[lvl];
          o = roi_align_fwd_elem(reinterpret_cast<const TIn*>(a.L.data[lvl]) + ((long)(n / a.R) * a.C + c) * H * W,
                                 H, W, bx.x, bx.y, bx.z, bx.w, a.L.scale[lvl], g / POOL, g % POOL, POOL, POOL);
        }
        if (a.L.nlvl > 1) o.val = o.val + 0.0f;
        if constexpr (HALF) reinterpret_cast<__half*>(a.out)[((long)n * a.C + c) * PPG + g] = __float2half(o.val);
        else a.out[((long)n * a.C + c) * PPG + g] = o.val;
        if (PK) {
          a.amax8[((long)n * a.C + c) * PPSG + g] = (unsigned char)o.code;
        } else {
          a.ax[((long)n * a.C + c) * PPG + g] = o.ax;
          a.ay[((long)n * a.C + c) * PPG + g] = o.ay;
        }
      }
      }  // flagged RoIs
    }
    }
#ifdef SD_PROFILING
  if (a.dbg && lane == 0) {
    long long* d = a.dbg + ((long)blockIdx.x * kBandWaves + wave) * 8;
    d[0] = t_setup; d[1] = t_wait; d[2] = t_comp; d[3] = dbg_count;
    d[4] = __builtin_readcyclecounter() - t_begin; d[5] = t_begin - t_entry; d[6] = dbg_units; d[7] = t_begin;
  }
#endif
}

// ------------------------------------------------------------------------------------------------
// Whole-plane forward of the drop-in ROIAlign_v2 (round 5): the C4 family -- a single level whose four
// channel planes fit in LDS together ((2,1024,50,84): 67 KB), 7x7 bins, float arg-max outputs.  The dual of
// roi_align_bwd_flt4_kernel: workgroup = (channel quad, image), two per CU.
//   planes   staged CHANNEL-LAST ([position][4 channels]: a 4 x 4 register block per lane, four 16-byte LDS
//            stores); a tap of four channels is one ds_read_b128, and the sample positions / fractions of a
//            (RoI, bin) -- the pre-pass's row / column entries, the ones the band kernel keeps in registers --
//            are shared by the four channels (the arithmetic runs on channel PAIRS: v_pk_mul / v_pk_add).
//   stores   92 % of this op's traffic is its three fp32 outputs.  Written per lane as 4-byte elements they
//            reach 2.9 TB/s (twelve 256-byte pieces per wave and trip; `profiles/r05r_*`: the stores ALONE
//            take 0.21 ms, a plain fill of the same 617 MB 0.09).  So a wave owns one RoI per trip (lane =
//            bin, 49 of 64 lanes), transposes each output through 784 bytes of wave-private LDS and writes the
//            784-byte run of (RoI, four channels) as 49 x 16 bytes: three stores per trip instead of twelve,
//            whole aligned runs.  (LDS operations of one wave complete in order: no barrier.)
// Same float expressions in the same order as roi_align_fwd_band (and roi_align_fwd_elem): bit-equal.
// RoIs the pre-pass flags (three-sample bins, nothing pooled) take roi_align_fwd_elem / the constant.
//   grid: x = channel quad, y = image; LDS = (H * W + 1) * 16 bytes + 8 x 784
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void roi_align_fwd_quad(BandArgs A) {
  extern __shared__ __attribute__((aligned(16))) float band_smem[];
  const FwdArgs& a = A.f;
  const BandPlan& P = A.p;
  constexpr int T = 512, POOL = 7, PPG = POOL * POOL, NW = T / kWave;
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  const int G = a.C / 4;
  const int c = 4 * ((G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x);
  const int img = blockIdx.y;
  const int H = a.L.H[0], W = a.L.W[0], HW = H * W;
  float4* pl = reinterpret_cast<float4*>(band_smem);
  float* st = band_smem + 4 * (HW + 1) + wave * (4 * PPG);   // this wave's 784 bytes
  const float* src = reinterpret_cast<const float*>(a.L.data[0]) + ((long)img * a.C + c) * HW;
  if ((HW & 3) == 0) {
    for (int i4 = tid; i4 < (HW >> 2); i4 += T) {
      const float4 v0 = *reinterpret_cast<const float4*>(src + 4 * i4);
      const float4 v1 = *reinterpret_cast<const float4*>(src + (long)HW + 4 * i4);
      const float4 v2 = *reinterpret_cast<const float4*>(src + 2L * HW + 4 * i4);
      const float4 v3 = *reinterpret_cast<const float4*>(src + 3L * HW + 4 * i4);
      pl[4 * i4 + 0] = make_float4(v0.x, v1.x, v2.x, v3.x);
      pl[4 * i4 + 1] = make_float4(v0.y, v1.y, v2.y, v3.y);
      pl[4 * i4 + 2] = make_float4(v0.z, v1.z, v2.z, v3.z);
      pl[4 * i4 + 3] = make_float4(v0.w, v1.w, v2.w, v3.w);
    }
  } else {   // planes that are not whole 16-byte words (P5 of 800 x 1333: 25 x 42): element loads
    for (int i = tid; i < HW; i += T)
      pl[i] = make_float4(src[i], src[(long)HW + i], src[2L * HW + i], src[3L * HW + i]);
  }
  if (tid == 0) pl[HW] = make_float4(0.f, 0.f, 0.f, 0.f);   // the "right" tap read beside the last pixel (weight 0 or replaced)
  __syncthreads();
  // lanes 49..63 repeat lane 48's work (same entries, same values, same addresses): every LDS and global store below is
  // issued by all 64 lanes, no branch around any of them
  const bool act = lane < PPG;
  const int bin = act ? lane : PPG - 1, pp = bin / POOL, q = bin - pp * POOL;
  // A software pipeline over the wave's RoIs, built around vmcnt: it counts loads and stores in order, so a wait for
  // table entries is also a wait for every store issued before their loads.  Per trip: (1) the loads of the NEXT
  // RoI's entries go out, (2) the three 784-byte stores of the PREVIOUS RoI's outputs go out behind them, (3) this
  // RoI is computed, (4) `touch` waits for the entries with exactly those three stores still in flight --
  // s_waitcnt vmcnt(3), which the compiler can only emit because every trip issues the same memory operations: the
  // loads are unconditional (clamped RoI index), flagged RoIs store constants and the few that need
  // roi_align_fwd_elem are redone by a second walk, the first trip "stores" its own RoI's rows (overwritten by the
  // real values a trip later: same wave, same addresses, in order).  With the stores of a trip waited for at the next
  // loop head (vmcnt(0)) compute and stores added up: 0.07 + 0.095 ms (`profiles/r05r_*`).
  struct Ent { uint4 re, ce; float2 rv, cv; int flag; };
  auto load_ent = [&](int n, Ent& e) {
    const long roi = (long)img * a.R + (n < a.R ? n : a.R - 1);
    e.flag = P.fbflag[roi];
    e.re = P.rowent[roi * POOL + pp];
    e.ce = P.colent[roi * POOL + q];
    e.rv = P.rowval[roi * POOL + pp];
    e.cv = P.colval[roi * POOL + q];
  };
  auto touch = [](Ent& e) {
    asm volatile("" : "+v"(e.re.x), "+v"(e.re.y), "+v"(e.re.z), "+v"(e.re.w), "+v"(e.ce.x), "+v"(e.ce.y), "+v"(e.ce.z),
                 "+v"(e.ce.w), "+v"(e.rv.x), "+v"(e.rv.y), "+v"(e.cv.x), "+v"(e.cv.y), "+v"(e.flag));
  };
  // [channel][bin] through the wave's LDS row, then 16 bytes per lane = the 784-byte run of (RoI, c .. c + 3)
  const int lane_c = act ? lane : PPG - 1;
  auto put = [&](const float (&v)[4], float* dst, long ob) {
#pragma unroll
    for (int k = 0; k < 4; ++k) st[k * PPG + bin] = v[k];
    *reinterpret_cast<float4*>(dst + ob + 4 * lane_c) = *reinterpret_cast<const float4*>(st + 4 * lane_c);
  };
  Ent cur;
  load_ent(wave, cur);
  float pv[4] = {0.f, 0.f, 0.f, 0.f}, px[4] = {-1.f, -1.f, -1.f, -1.f}, py[4] = {-1.f, -1.f, -1.f, -1.f};
  long pob = (((long)img * a.R + (wave < a.R ? wave : 0)) * a.C + c) * PPG;   // (first trip: this wave's own first RoI)
  for (int n = wave; n < a.R; n += NW) {
    const long roi = (long)img * a.R + n;
    const int flag = __builtin_amdgcn_readfirstlane(cur.flag);
    const uint4 re = cur.re, ce = cur.ce;
    const float2 rv = cur.rv, cv = cur.cv;
    Ent nxt;
    load_ent(n + NW, nxt);
    if (!(SD_ABLATE(a, 128))) put(pv, a.out, pob);   // (profiling build: 128 no value stores, 256 no arg-max stores)
    if (!(SD_ABLATE(a, 256))) {
      put(px, a.ax, pob);
      put(py, a.ay, pob);
    }
    float maxval[4], bx_[4], by_[4];
    if (flag) {   // nothing pooled (flag 2): the constant; flag 1 is redone below
#pragma unroll
      for (int k = 0; k < 4; ++k) { maxval[k] = 0.f; bx_[k] = -1.f; by_[k] = -1.f; }
    } else {
      const int lo0 = (int)(re.x & 0xfffff), lo1 = (int)(re.y & 0xfffff);
      const int hi0 = lo0 + ((re.x >> 20) & 1 ? W : 0), hi1 = lo1 + ((re.y >> 20) & 1 ? W : 0);
      const int left0 = ce.x & 0xfff, left1 = (ce.x >> 13) & 0xfff;
      // coincident (left, right) columns: both taps are the left pixel
      const int right0 = left0 + ((ce.x >> 12) & 1 ? 0 : 1), right1 = left1 + ((ce.x >> 25) & 1 ? 0 : 1);
      const bool empty = (re.x >> 31) | ((ce.x >> 26) & 1);
      const float fa0 = __uint_as_float(re.z), fa1 = __uint_as_float(re.w);
      const float fb0 = __uint_as_float(ce.z), fb1 = __uint_as_float(ce.w);
      // weight products, the reference's expressions (roi_align_v2-inl.h:131-134), paired (left, right)
      const v2f b0 = {1 - fb0, fb0}, b1 = {1 - fb1, fb1};
      const v2f wl00 = (1 - fa0) * b0, wh00 = fa0 * b0, wl01 = (1 - fa0) * b1, wh01 = fa0 * b1;
      const v2f wl10 = (1 - fa1) * b0, wh10 = fa1 * b0, wl11 = (1 - fa1) * b1, wh11 = fa1 * b1;
      const float init = empty ? 0.f : -FLT_MAX;
#pragma unroll
      for (int k = 0; k < 4; ++k) { maxval[k] = init; bx_[k] = -1.f; by_[k] = -1.f; }
      // one sample: rows (lo, hi), columns (left, right), four channels at a time as two channel pairs;
      // value = w1*TL + w2*BL + w3*TR + w4*BR, summed left to right (roi_align_v2-inl.h:135-138)
      auto sample = [&](int lo, int hi, int left, int right, v2f wl, v2f wh, float cx, float cy) {
        const float4 tl = pl[lo + left], bl = pl[hi + left], tr = pl[lo + right], br = pl[hi + right];
        const v2f w1 = {wl.x, wl.x}, w2 = {wh.x, wh.x}, w3 = {wl.y, wl.y}, w4 = {wh.y, wh.y};
        const v2f va = ((w1 * v2f{tl.x, tl.y} + w2 * v2f{bl.x, bl.y}) + w3 * v2f{tr.x, tr.y}) + w4 * v2f{br.x, br.y};
        const v2f vb = ((w1 * v2f{tl.z, tl.w} + w2 * v2f{bl.z, bl.w}) + w3 * v2f{tr.z, tr.w}) + w4 * v2f{br.z, br.w};
        const float value[4] = {va.x, va.y, vb.x, vb.y};
#pragma unroll
        for (int k = 0; k < 4; ++k)
          if (value[k] > maxval[k]) { maxval[k] = value[k]; bx_[k] = cx; by_[k] = cy; }
      };
      sample(lo0, hi0, left0, right0, wl00, wh00, cv.x, rv.x);
      sample(lo0, hi0, left1, right1, wl01, wh01, cv.y, rv.x);
      sample(lo1, hi1, left0, right0, wl10, wh10, cv.x, rv.y);
      sample(lo1, hi1, left1, right1, wl11, wh11, cv.y, rv.y);
    }
    touch(nxt);
    cur = nxt;
#pragma unroll
    for (int k = 0; k < 4; ++k) { pv[k] = maxval[k]; px[k] = bx_[k]; py[k] = by_[k]; }
    pob = (roi * a.C + c) * PPG;
  }
  if (wave < a.R) {   // the last RoI's outputs
    if (!(SD_ABLATE(a, 128))) put(pv, a.out, pob);
    if (!(SD_ABLATE(a, 256))) {
      put(px, a.ax, pob);
      put(py, a.ay, pob);
    }
  }
  // the RoIs of the exact per-element path (three-sample bins: a handful per launch): element stores over the
  // constants written above (same wave, same addresses, in order)
  for (int n = wave; n < a.R; n += NW) {
    const long roi = (long)img * a.R + n;
    if (P.fbflag[roi] != 1) continue;
    const float4 bx = *reinterpret_cast<const float4*>(a.rois + roi * 4);
    const long ob = (roi * a.C + c) * PPG;
#pragma unroll 1
    for (int k = 0; k < 4; ++k) {
      if (!act) continue;
      const FwdOut o = roi_align_fwd_elem(src + (long)k * HW, H, W, bx.x, bx.y, bx.z, bx.w, a.L.scale[0], pp, q, POOL, POOL);
      a.out[ob + k * PPG + bin] = o.val;
      a.ax[ob + k * PPG + bin] = o.ax;
      a.ay[ob + k * PPG + bin] = o.ay;
    }
  }
}


__global__ __launch_bounds__(256) void fpn_assign_kernel(const float* rois, int n_rois,
                                                         RoiLevels L, float* rois_per_level,
                                                         int32_t* level) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_rois) return;
  const float* r = rois + (long)i * 4;
  const float x1 = r[0], y1 = r[1], x2 = r[2], y2 = r[3];
  const int lvl = fpn_level(x1, y1, x2, y2, L);
  if (level) level[i] = lvl;
  if (rois_per_level)
    for (int l = 0; l < L.nlvl; ++l) {
      float4 v = (l == lvl) ? make_float4(x1, y1, x2, y2) : make_float4(0.f, 0.f, 0.f, 0.f);
      *reinterpret_cast<float4*>(rois_per_level + ((long)l * n_rois + i) * 4) = v;
    }
}

// ------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------
int fill_levels(RoiLevels& L, const float* const* feats, const int* Hs, const int* Ws,
                       const int* strides, int nlvl, float canon_scale, float canon_level) {
  SD_REQUIRE(nlvl >= 1 && nlvl <= SD_MAX_FPN_LEVELS, "nlvl=%d out of range [1,%d]", nlvl,
             SD_MAX_FPN_LEVELS);
  int smin = strides[0], smax = strides[0];
  for (int l = 0; l < nlvl; ++l) {
    SD_REQUIRE(Hs[l] > 0 && Ws[l] > 0 && strides[l] > 0, "level %d: bad H/W/stride", l);
    L.data[l] = feats ? feats[l] : nullptr;
    L.H[l] = Hs[l];
    L.W[l] = Ws[l];
    L.stride[l] = strides[l];
    L.scale[l] = 1.0f / (float)strides[l];
    if (strides[l] < smin) smin = strides[l];
    if (strides[l] > smax) smax = strides[l];
  }
  L.nlvl = nlvl;
  L.canon_scale = canon_scale;
  L.canon_level = canon_level;
  L.k_min = (float)log2((double)smin);
  L.k_max = (float)log2((double)smax);
  return SD_OK;
}

// bplan: a backward plan (launch_bwd_fused prepass = 1) whose list / tap-table pre-pass is to run in
// the forward's pre-pass launch; *bplan_done tells whether it did (band-resident path only).
int launch_fwd(FwdArgs& a, hipStream_t st, void* workspace, size_t workspace_bytes, const BwdFusedArgs* bplan,
               bool* bplan_done) {
  if (bplan_done) *bplan_done = false;
  const long count = (long)a.B * a.R * a.C * a.PH * a.PW;
  if (count == 0) return SD_OK;
  // 0: the naive per-element kernel only, 1 (default): band-resident kernel, tiled kernels where it does not apply
  const int variant = tuning("roi_align_fwd", 1);
  a.ablate = SD_PROF_TUNING("roi_align_fwd_ablate", 0);
  a.dbg = reinterpret_cast<long long*>(((uintptr_t)(unsigned)SD_PROF_TUNING("roi_align_dbg_hi", 0) << 32) |
                                       (uintptr_t)(unsigned)SD_PROF_TUNING("roi_align_dbg_lo", 0));
  const int nroi = a.B * a.R;
  // tiled fallback: 8 channel slices (workgroups) per RoI -- the largest divisor of C not above that
  a.nslice = 1;
  for (int d = 1; d <= a.C && d <= 8; ++d)
    if (a.C % d == 0) a.nslice = d;
  // the band kernel's exact path (a handful of RoIs per launch, after the band work): 32 slices -- with 8
  // the few workgroups that own a flagged RoI finish 1.5-2 us after everybody else (same-box A/B, 4 pairs)
  // (single-level calls keep 8: a per-level op of the unfused FPN graph sees three quarters of its RoIs as
  // "void" rows of the exact path, which then wants fuller workgroups: 63.6 -> 81.8 us with 32)
  const int wantfb = a.L.nlvl > 1 ? 32 : 8;
  a.fbslice = 1;
  for (int d = 1; d <= a.C && d <= wantfb; ++d)
    if (a.C % d == 0) a.fbslice = d;
  // the tiled kernels fetch (left,right) column pairs with one 8-byte load: needs W >= 2
  bool wide = true;
  for (int l = 0; l < a.L.nlvl; ++l)
    if (a.L.stride[l] >= 0 && a.L.W[l] < 2) wide = false;
  // ---- band-resident forward (default): pre-pass + one launch; needs the workspace ----
  if (variant == 1 && wide && ((a.PH == 7 && a.PW == 7) || (a.PH == 14 && a.PW == 14)) &&
      a.R <= 65535 && workspace && tuning("roi_align_fwd_band", 1)) {
    BandArgs A{};
    BandPlan& P = A.p;
    const int POOL = a.PH;
    bool ok = true;
    int units = 0;
    constexpr int gmax = 8;   // most planes per fill
    int nvalid_lv = 0;
    for (int l = 0; l < a.L.nlvl; ++l) nvalid_lv += a.L.stride[l] >= 0;
    for (int l = 0; l < a.L.nlvl; ++l) {
      if (a.L.stride[l] < 0) continue;
      const int H = a.L.H[l], W = a.L.W[l];
      const long HW = (long)H * W;
      if (W > 4095 || HW >= (1 << 20)) { ok = false; break; }
      // halo: a bin row of a RoI that covers the whole map taps ceil(H / POOL) + 2 rows; small
      // maps (P3..P5, the C4 map) get a halo that makes every RoI eligible, the finest level
      // keeps 8 rows (its RoIs are small by the FPN assignment; the rest takes the exact path)
      int halo = kBandHalo;
      if (H <= 128) {
        halo = (H + POOL - 1) / POOL + 2;
        halo = halo < kBandHalo ? kBandHalo : (halo > 12 ? 12 : halo);
      }
      // bands: as few as LDS allows, but enough that a unit's expected items (an even share of the
      // image's R * POOL bin rows per level) fit one round of the workgroup
      const int rb = (kBandBufFloats - 16) / W;  // rows one buffer holds
      if (rb < halo + 4) { ok = false; break; }
      int nbn = H <= rb ? 1 : (H + (rb - halo) - 1) / (rb - halo);
      const int cap = kBandNP * kBandWaves * (kWave / POOL);
      const long est = (long)a.R * POOL / (nvalid_lv > 0 ? nvalid_lv : 1);
      const int by_items = (int)((est * 5 + 4L * cap - 1) / (4L * cap));   // est / (0.8 cap)
      // (packed arg-max only: with the three fp32 outputs of the drop-in op the stores dominate, and
      // a RoI whose bin rows sit in one band is written as whole 196-byte rows -- C4: 254 vs 334 us;
      // the extra rounds re-read planes that are still in L2)
      if (by_items > nbn && a.amax8) nbn = by_items;
      if (nbn > H) nbn = H;
      if (nbn > kBandMaxBands) nbn = kBandMaxBands;   // (more items than that: rounds)
      int owned = (H + nbn - 1) / nbn;
      if (nbn > 1)
        for (int o = owned; o < owned + 4 && o + halo <= rb; ++o)
          if (((long)o * W) % 4 == 0) { owned = o; break; }  // 16-byte aligned band starts
      if (owned + halo > rb) owned = rb - halo;
      nbn = (H + owned - 1) / owned;
      if (nbn > kBandMaxBands) { ok = false; break; }
      P.nbands[l] = nbn;
      P.halo[l] = halo;
      P.owned[l] = nbn == 1 ? H : owned;
      P.rows[l] = nbn == 1 ? H : (owned + halo < H ? owned + halo : H);
      const long bstride = a.half_io ? (((long)P.rows[l] * W + 14) >> 3) * 8 : (((long)P.rows[l] * W + 4 + 3) & ~3L);
      int g = 1;
      for (int c = 2; c <= 8 && c <= gmax; c *= 2)
        if (a.C % c == 0 && c * bstride <= kBandBufFloats) g = c;
      if (bstride > kBandBufFloats) { ok = false; break; }
      P.g[l] = g;
      P.unit_base[l] = units;
      units += a.B * P.nbands[l];
    }
    // workspace carve-up
    auto al16 = [](size_t x) { return (x + 15) & ~(size_t)15; };
    const size_t ent = al16((size_t)nroi * POOL * sizeof(uint4));
    const size_t valb = a.amax8 ? 0 : al16((size_t)nroi * POOL * sizeof(float2));
    const size_t itemb = al16((size_t)a.B * SD_MAX_FPN_LEVELS * kBandSub * ((a.R + kBandSub - 1) / kBandSub) *
                              POOL * sizeof(unsigned));
    const size_t segb = al16((size_t)units * kBandSub * sizeof(int2));
    const size_t need = 16 + 2 * ent + 2 * valb + itemb + segb + al16(nroi) + kBandMaxUnits * sizeof(int);
    int wg = 0;
    P.nunits = units;
    // channels a workgroup reserves at a time: 4 (single-channel grabs +7 %: the 196-byte rows of
    // neighbouring channels merge in one CU's L2); no cost bias for multi-plane units and no shrinking
    // reservations at a unit's end (15 / 30 % and 1..8-plane tails: measured, no effect -- round 3)
    P.grab = 4;
    P.gbias = 0;
    P.tail = 0;
    P.tail_planes = 8;
    {
      // virtual units = sum over units of ceil(items / CAP) <= units + floor(all items / CAP), and a RoI
      // has at most POOL items: beyond the table's size the launch goes to the tiled kernels (the
      // kernel's table holds kBandMaxUnits entries and drops nothing below that)
      const int np = (a.amax8 && !a.half_io) ? kBandNP : kBandNP - 1;
      const long cap_launch = (long)np * kBandWaves * (kWave / POOL);
      if (units + (long)nroi * POOL / cap_launch > kBandMaxUnits) ok = false;
    }
    wg = kNumCU;  // persistent workgroups, one per CU
    if (ok && need <= workspace_bytes) {
      char* w = reinterpret_cast<char*>(((uintptr_t)workspace + 15) & ~(uintptr_t)15);
      P.rowent = reinterpret_cast<uint4*>(w); w += ent;
      P.colent = reinterpret_cast<uint4*>(w); w += ent;
      if (valb) {
        P.rowval = reinterpret_cast<float2*>(w); w += valb;
        P.colval = reinterpret_cast<float2*>(w); w += valb;
      }
      P.items = reinterpret_cast<unsigned*>(w); w += itemb;
      P.seg = reinterpret_cast<int2*>(w); w += segb;
      P.fbflag = reinterpret_cast<unsigned char*>(w); w += al16(nroi);
      P.chan_ctr = reinterpret_cast<int*>(w);
      P.nwg = wg;
      P.pool = POOL;
      A.f = a;
      const int nlist = a.B * a.L.nlvl * kBandSub, nent = cdiv((long)nroi * 2 * POOL, kBandThreads);
      const int ncoord = a.amax8 ? cdiv((long)nroi * 6 * POOL, kBandThreads) : 0;
      P.nlist = nlist; P.nent = nent;
      const int smem = 2 * kBandBufFloats * (int)sizeof(float);
      const bool merged = bplan && bplan->lists_units > 0 && bplan->PP == POOL * POOL;
      if (bplan_done) *bplan_done = merged;
      if (int e = launch_fwd_prep(A, POOL, nlist + nent + ncoord, merged ? bplan : nullptr, st)) return e;
      // single level, 7x7, float arg-max, four whole planes fit in LDS together (the C4 family): one workgroup
      // per (channel quad, image) with channel-last planes and 784-byte stores (`roi_align_fwd_quad` = 0: the band kernel)
      {
        const long HW0 = (long)a.L.H[0] * a.L.W[0];
        const size_t lds = (size_t)(HW0 + 1) * 16 + 8 * 4 * 49 * sizeof(float);
        if (!a.amax8 && !a.half_io && POOL == 7 && a.L.nlvl == 1 && a.L.stride[0] >= 0 && P.nbands[0] == 1 &&
            a.C % 4 == 0 && lds <= 78 * 1024 && a.B <= 65535 &&
            (((uintptr_t)a.L.data[0] | (uintptr_t)a.out | (uintptr_t)a.ax | (uintptr_t)a.ay) & 15) == 0 &&
            tuning("roi_align_fwd_quad", 1) == 1) {
          SD_HIP_CHECK(hipFuncSetAttribute((const void*)roi_align_fwd_quad, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          hipLaunchKernelGGL(roi_align_fwd_quad, dim3(a.C / 4, a.B), dim3(512), lds, st, A);
          note_dispatch("sd::roi_fwd_prep_kernel<7> + sd::roi_align_fwd_quad");
          SD_LAUNCH_CHECK();
          return SD_OK;
        }
      }
#define SD_FWD_BAND(POOLV, PK)                                                                    \
  do {                                                                                            \
    auto k = roi_align_fwd_band<POOLV, PK>;                                                       \
    SD_HIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                     smem));                                                      \
    hipLaunchKernelGGL(k, dim3(wg + kBandFallbackWGs), dim3(kBandThreads), smem, st, A);          \
  } while (0)
      if (a.half_io) {  // fp16 features and output, packed arg-max
#define SD_FWD_BAND_H(POOLV)                                                                      \
  do {                                                                                            \
    auto k = roi_align_fwd_band<POOLV, true, true>;                                               \
    SD_HIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize,  \
                                     smem));                                                      \
    hipLaunchKernelGGL(k, dim3(wg + kBandFallbackWGs), dim3(kBandThreads), smem, st, A);          \
  } while (0)
        if (POOL == 7) SD_FWD_BAND_H(7); else SD_FWD_BAND_H(14);
#undef SD_FWD_BAND_H
      } else if (POOL == 7) {
        if (a.amax8) SD_FWD_BAND(7, true); else SD_FWD_BAND(7, false);
      } else {
        if (a.amax8) SD_FWD_BAND(14, true); else SD_FWD_BAND(14, false);
      }
#undef SD_FWD_BAND
      note_dispatch("sd::%s<%d> + sd::roi_align_fwd_band<%d,%s,%s>", merged ? "roi_prep_merged_kernel" : "roi_fwd_prep_kernel",
                    POOL, POOL, a.amax8 ? "true" : "false", a.half_io ? "true" : "false");
      SD_LAUNCH_CHECK();
      return SD_OK;
    }
  }
  if (a.half_io)
    return fail(SD_ERR_UNSUPPORTED, "fp16 RoIAlign runs on the band-resident kernel only: it needs the workspace, "
                "7x7 or 14x14 pooling, W in [2, 4095] and roi_align_fwd = 1, roi_align_fwd_band = 1");
  note_dispatch("sd::roi_align_fwd (tiled / naive fallback kernels: no workspace or a shape the band kernel does not take)");
  if (variant >= 1 && wide && a.PH == 7 && a.PW == 7) {
    // four RoIs per workgroup share one set of axis tables
    if (a.amax8)
      hipLaunchKernelGGL((roi_align_fwd_tiled<7, 7, 4, true>), dim3(cdiv(nroi, 4) * a.nslice), dim3(512), 0, st, a);
    else
      hipLaunchKernelGGL((roi_align_fwd_tiled<7, 7, 4, false>), dim3(cdiv(nroi, 4) * a.nslice), dim3(512), 0, st, a);
  } else if (variant >= 1 && wide && a.PH == 14 && a.PW == 14) {
    if (a.amax8)
      hipLaunchKernelGGL((roi_align_fwd_tiled<14, 14, 1, true>), dim3(nroi * a.nslice), dim3(512), 0,
                         st, a);
    else
      hipLaunchKernelGGL((roi_align_fwd_tiled<14, 14, 1, false>), dim3(nroi * a.nslice), dim3(512),
                         0, st, a);
  } else {
    const int grid = (int)((count + 255) / 256 < 65536 * 16 ? (count + 255) / 256 : 65536 * 16);
    hipLaunchKernelGGL(roi_align_fwd_naive, dim3(grid), dim3(256), 0, st, a);
    if (a.amax8)  // the tiled kernels write the coordinate table themselves
      hipLaunchKernelGGL(roi_coords_kernel, dim3(nroi), dim3(64), 0, st, a.rois, nroi, a.L, a.PH,
                         a.PW, a.coords);
  }
  SD_LAUNCH_CHECK();
  return SD_OK;
}

int check_dims(int B, int C, int R, int ph, int pw) {
  SD_REQUIRE(B >= 0 && C >= 0 && R >= 0, "negative dimension (B=%d C=%d R=%d)", B, C, R);
  SD_REQUIRE(ph > 0 && pw > 0, "pooled_size must be nonzero (got %d x %d)", ph, pw);
  SD_REQUIRE((long)B * R * C * ph * pw < (1L << 31), "output has >= 2^31 elements");
  return SD_OK;
}

}  // namespace sd

using namespace sd;

extern "C" int sd_roi_align_v2_fwd(const float* data, const float* rois, float* out,
                                   float* maxidx_x, float* maxidx_y, int B, int C, int H, int W,
                                   int R, int pooled_h, int pooled_w, float spatial_scale,
                                   void* stream) {
  return sd_roi_align_v2_fwd_ws(data, rois, out, maxidx_x, maxidx_y, B, C, H, W, R, pooled_h, pooled_w,
                                spatial_scale, nullptr, 0, stream);
}

extern "C" size_t sd_roi_align_v2_workspace_bytes(int B, int R) {
  return sd_fpn_roi_align_workspace_bytes(B, R);
}

extern "C" int sd_roi_align_v2_fwd_ws(const float* data, const float* rois, float* out,
                                      float* maxidx_x, float* maxidx_y, int B, int C, int H, int W,
                                      int R, int pooled_h, int pooled_w, float spatial_scale,
                                      void* workspace, size_t workspace_bytes, void* stream) {
  if (int e = check_dims(B, C, R, pooled_h, pooled_w)) return e;
  SD_REQUIRE(H > 0 && W > 0 && (long)H * W < (1L << 30), "bad feature size %d x %d", H, W);
  SD_REQUIRE(spatial_scale >= 0.f && spatial_scale <= 1.f, "spatial_scale %g outside [0,1]",
             (double)spatial_scale);
  SD_REQUIRE((data && rois && out && maxidx_x && maxidx_y) || (long)B * R * C == 0,
             "null tensor pointer");
  FwdArgs a{};
  a.L.nlvl = 1;
  a.L.data[0] = data;
  a.L.H[0] = H;
  a.L.W[0] = W;
  a.L.stride[0] = 0;
  a.L.scale[0] = spatial_scale;
  a.rois = rois; a.out = out; a.ax = maxidx_x; a.ay = maxidx_y;
  a.B = B; a.C = C; a.R = R; a.PH = pooled_h; a.PW = pooled_w;
  return launch_fwd(a, (hipStream_t)stream, workspace, workspace_bytes);
}

extern "C" int sd_fpn_roi_align_fwd(const float* const* feats_host, const int* Hs_host,
                                    const int* Ws_host, const int* strides_host, int nlvl,
                                    const float* rois, float* out, float* maxidx_x,
                                    float* maxidx_y, int B, int C, int R, int pooled_h,
                                    int pooled_w, float roi_canonical_scale,
                                    float roi_canonical_level, void* workspace,
                                    size_t workspace_bytes, void* stream) {
  if (int e = check_dims(B, C, R, pooled_h, pooled_w)) return e;
  SD_REQUIRE(feats_host && Hs_host && Ws_host && strides_host, "null level description");
  FwdArgs a{};
  if (int e = fill_levels(a.L, feats_host, Hs_host, Ws_host, strides_host, nlvl,
                          roi_canonical_scale, roi_canonical_level))
    return e;
  for (int l = 0; l < nlvl; ++l) SD_REQUIRE(feats_host[l] || (long)B * C == 0, "feats[%d] null", l);
  if (nlvl == 1) a.L.nlvl = 2, a.L.stride[1] = -1;  // keep the assignment filter on (1-level FPN)
  a.rois = rois; a.out = out; a.ax = maxidx_x; a.ay = maxidx_y;
  a.B = B; a.C = C; a.R = R; a.PH = pooled_h; a.PW = pooled_w;
  return launch_fwd(a, (hipStream_t)stream, workspace, workspace_bytes);
}

extern "C" int sd_fpn_roi_align_argmax_stride(int pooled_h, int pooled_w) {
  return amax_stride(pooled_h * pooled_w);
}

static int fpn_fwd_packed_impl(const float* const* feats_host, const int* Hs_host, const int* Ws_host,
                               const int* strides_host, int nlvl, const float* rois, float* out,
                               uint8_t* argmax, float* coords, int B, int C, int R, int pooled_h,
                               int pooled_w, float roi_canonical_scale, float roi_canonical_level,
                               void* workspace, size_t workspace_bytes, void* plan, size_t plan_bytes,
                               void* stream) {
  if (int e = check_dims(B, C, R, pooled_h, pooled_w)) return e;
  SD_REQUIRE(feats_host && Hs_host && Ws_host && strides_host, "null level description");
  SD_REQUIRE((argmax && coords) || (long)B * R * C == 0, "argmax / coords is null");
  SD_REQUIRE(((uintptr_t)argmax & 3) == 0 && ((uintptr_t)coords & 7) == 0,
             "argmax must be 4-byte and coords 8-byte aligned");
  FwdArgs a{};
  if (int e = fill_levels(a.L, feats_host, Hs_host, Ws_host, strides_host, nlvl,
                          roi_canonical_scale, roi_canonical_level))
    return e;
  for (int l = 0; l < nlvl; ++l) SD_REQUIRE(feats_host[l] || (long)B * C == 0, "feats[%d] null", l);
  if (nlvl == 1) a.L.nlvl = 2, a.L.stride[1] = -1;
  a.rois = rois; a.out = out; a.amax8 = argmax; a.coords = coords;
  a.B = B; a.C = C; a.R = R; a.PH = pooled_h; a.PW = pooled_w;
  hipStream_t st = (hipStream_t)stream;
  // the backward's list / tap-table pre-pass rides in the forward's pre-pass launch when the caller
  // hands over the plan buffer the backward will read (sd_fpn_roi_align_bwd_packed_plan)
  BwdFusedArgs f{};
  bool have = false;
  if (plan && ((uintptr_t)plan & 15) == 0 && (long)B * R * C > 0 && R <= 8192 &&
      ((pooled_h == 7 && pooled_w == 7) || (pooled_h == 14 && pooled_w == 14))) {
    f.L = a.L;
    f.amax8 = argmax; f.coords = coords; f.rois = rois;
    for (int l = 0; l < nlvl; ++l) f.dx[l] = reinterpret_cast<float*>(uintptr_t(16));  // (planning only: "wanted")
    f.B = B; f.C = C; f.R = R; f.PP = pooled_h * pooled_w; f.filter = 1; f.req = SD_REQ_WRITE;
    have = launch_bwd_fused(f, nlvl, st, plan, plan_bytes, 1) == SD_OK;
  }
  bool done = false;
  if (int e = launch_fwd(a, st, workspace, workspace_bytes, have ? &f : nullptr, &done)) return e;
  if (have && !done)   // the forward ran on a fallback kernel: the stand-alone list pre-pass
    if (int e = launch_bwd_lists(f, f.lists_units, st)) return e;
  return SD_OK;
}

extern "C" int sd_fpn_roi_align_fwd_packed(const float* const* feats_host, const int* Hs_host,
                                           const int* Ws_host, const int* strides_host, int nlvl,
                                           const float* rois, float* out, uint8_t* argmax,
                                           float* coords, int B, int C, int R, int pooled_h,
                                           int pooled_w,
                                           float roi_canonical_scale, float roi_canonical_level,
                                           void* workspace, size_t workspace_bytes, void* stream) {
  return fpn_fwd_packed_impl(feats_host, Hs_host, Ws_host, strides_host, nlvl, rois, out, argmax, coords, B, C,
                             R, pooled_h, pooled_w, roi_canonical_scale, roi_canonical_level, workspace,
                             workspace_bytes, nullptr, 0, stream);
}

extern "C" int sd_fpn_roi_align_fwd_packed_plan(const float* const* feats_host, const int* Hs_host,
                                                const int* Ws_host, const int* strides_host, int nlvl,
                                                const float* rois, float* out, uint8_t* argmax,
                                                float* coords, int B, int C, int R, int pooled_h,
                                                int pooled_w, float roi_canonical_scale,
                                                float roi_canonical_level, void* workspace,
                                                size_t workspace_bytes, void* plan, size_t plan_bytes,
                                                void* stream) {
  SD_REQUIRE(plan, "plan is null (use sd_fpn_roi_align_fwd_packed)");
  SD_REQUIRE(((uintptr_t)plan & 15) == 0, "plan must be 16-byte aligned");
  return fpn_fwd_packed_impl(feats_host, Hs_host, Ws_host, strides_host, nlvl, rois, out, argmax, coords, B, C,
                             R, pooled_h, pooled_w, roi_canonical_scale, roi_canonical_level, workspace,
                             workspace_bytes, plan, plan_bytes, stream);
}

extern "C" int sd_fpn_roi_align_fwd_packed_f16(const void* const* feats_host, const int* Hs_host,
                                               const int* Ws_host, const int* strides_host, int nlvl,
                                               const float* rois, void* out, uint8_t* argmax,
                                               float* coords, int B, int C, int R, int pooled_h,
                                               int pooled_w, float roi_canonical_scale,
                                               float roi_canonical_level, void* workspace,
                                               size_t workspace_bytes, void* stream) {
  if (int e = check_dims(B, C, R, pooled_h, pooled_w)) return e;
  SD_REQUIRE(feats_host && Hs_host && Ws_host && strides_host, "null level description");
  SD_REQUIRE((argmax && coords && out) || (long)B * R * C == 0, "out / argmax / coords is null");
  SD_REQUIRE(((uintptr_t)argmax & 3) == 0 && ((uintptr_t)coords & 7) == 0 && ((uintptr_t)out & 1) == 0,
             "argmax must be 4-byte, coords 8-byte and out 2-byte aligned");
  FwdArgs a{};
  if (int e = fill_levels(a.L, reinterpret_cast<const float* const*>(feats_host), Hs_host, Ws_host,
                          strides_host, nlvl, roi_canonical_scale, roi_canonical_level))
    return e;
  for (int l = 0; l < nlvl; ++l) {
    SD_REQUIRE(feats_host[l] || (long)B * C == 0, "feats[%d] null", l);
    SD_REQUIRE(((uintptr_t)feats_host[l] & 15) == 0, "feats[%d] must be 16-byte aligned", l);
  }
  if (nlvl == 1) a.L.nlvl = 2, a.L.stride[1] = -1;
  a.rois = rois; a.out = reinterpret_cast<float*>(out); a.amax8 = argmax; a.coords = coords;
  a.B = B; a.C = C; a.R = R; a.PH = pooled_h; a.PW = pooled_w;
  a.half_io = 1;
  return launch_fwd(a, (hipStream_t)stream, workspace, workspace_bytes);
}

extern "C" size_t sd_fpn_roi_align_workspace_bytes(int B, int R) {
  // band-resident forward: two 16-byte entries + two 8-byte coordinate pairs per (RoI, axis bin) of
  // the larger pooled size (14), the item lists of up to SD_MAX_FPN_LEVELS levels, unit segments
  // (<= kBandMaxBands bands per level), one flag byte per RoI
  const size_t b = B > 0 ? B : 0, r = R > 0 ? R : 0, nroi = b * r;
  return nroi * 14 * (2 * 16 + 2 * 8) + b * SD_MAX_FPN_LEVELS * (r + kBandSub) * 14 * 4 +
         b * SD_MAX_FPN_LEVELS * kBandMaxBands * kBandSub * 8 + nroi + kBandMaxUnits * 4 + 256;
}

extern "C" int sd_fpn_roi_assign(const float* rois, int n_rois, const int* strides_host, int nlvl,
                                 float roi_canonical_scale, float roi_canonical_level,
                                 float* rois_per_level, int32_t* level, void* stream) {
  SD_REQUIRE(n_rois >= 0, "n_rois < 0");
  SD_REQUIRE(strides_host, "strides null");
  RoiLevels L{};
  int ones[SD_MAX_FPN_LEVELS];
  for (int l = 0; l < SD_MAX_FPN_LEVELS; ++l) ones[l] = 1;
  if (int e = fill_levels(L, nullptr, ones, ones, strides_host, nlvl, roi_canonical_scale,
                          roi_canonical_level))
    return e;
  if (n_rois == 0) return SD_OK;
  SD_REQUIRE(rois, "rois null");
  hipLaunchKernelGGL(fpn_assign_kernel, dim3(cdiv(n_rois, 256)), dim3(256), 0,
                     (hipStream_t)stream, rois, n_rois, L, rois_per_level, level);
  SD_LAUNCH_CHECK();
  return SD_OK;
}

