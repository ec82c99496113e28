// DeformableConvolution forward WITHOUT a col matrix for gfx950: deformable sampling fused into the GEMM's
// B-operand staging (role-specialised 8-wave workgroups).
#include "deform_split.h"

namespace sd {

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

}  // namespace sd

using namespace sd;

// ---- forward without a col matrix (fused sampling + GEMM) ----------------------------------------
bool sd::dcn_fused_shape_ok(int C, int H, int W, int kh, int kw, int dgroup) {
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

int sd::deform_conv_fwd_nocol_impl(const float* x, const float* offset, const float* weight, const float* bias,
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
