// fp32-in / fp32-out matrix-core GEMM for gfx950 (the three products of the DeformableConvolution layer):
// exact fp32 products on v_mfma_f32_32x32x2_f32, or every product as three 16-bit MFMA terms of a hi / lo
// operand split (scaled fp16: the default; bf16: opt-in), plus the operand-maximum pre-pass the scaled
// split needs.  Replaces the linalg_gemm calls of deformable_convolution-inl.h (cuBLAS sgemm in the reference).
#include "deform_split.h"

namespace sd {

// ------------------------------------------------------------------------------------------------
// fp32 MFMA GEMM:  C[b] (M x N) (+)= A[b] (M x K) . B[b] (K x N)
//   element A(m,k) = A[m*sam + k*sak], B(k,n) = B[k*sbk + n*sbn]; one of each stride pair is 1.
// ------------------------------------------------------------------------------------------------

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
  int vec_store;         // split kernel: C rows are 16-byte aligned and N % 4 == 0 (plain stores leave as whole tile rows)
  int nt_store;          // split kernel: those stores are non-temporal
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

constexpr int SBM = 128, SBN = 128, SBK = 64;
constexpr int kSplitPlane = 128 * 128;  // bytes per LDS plane

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
  // Round 6: a plain store of a whole tile (C = A.B, the dcol product of the layer's backward: 619 MB of output for
  // four k steps per tile) goes through LDS -- written per lane as above every store instruction is two 128-byte
  // pieces of two rows that are not line aligned (16,800-byte rows), 64 of them per lane.  The accumulators are
  // transposed through the 64 KB the operand planes no longer need (column major: a lane's four consecutive rows are
  // one ds_write_b128; the 16-byte granule of (column c, rows 4 g ..) sits at g ^ swz(c), which keeps both the
  // writes' 8-lane groups and the reads' 16-lane groups on distinct banks), every thread takes 4 x 4 blocks back
  // (four ds_read_b128), and a store instruction writes 16 bytes per lane: two whole 512-byte tile rows.
  if (!piece && a.mode == 0 && a.vec_store) {
    __syncthreads();   // every wave is done with the operand planes
    auto swz = [](int c) { return ((c >> 2) ^ ((c & 3) << 2)) & 15; };
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int cc = wn + j * 32 + (lane & 31);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int r0 = wm + i * 32 + g * 8 + (lane >> 5) * 4;
          float4 v = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
          if (MODE == kSplitF16) {   // exact (powers of two)
            v.x = (v.x * inva) * invb; v.y = (v.y * inva) * invb; v.z = (v.z * inva) * invb; v.w = (v.w * inva) * invb;
          }
          *reinterpret_cast<float4*>(smem + cc * 512 + (((r0 >> 2) ^ swz(cc)) << 4)) = v;
        }
      }
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int blk = tid + it * 256, c0 = (blk & 31) * 4, r0 = (blk >> 5) * 4;
      float4 q[4];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        q[k] = *reinterpret_cast<const float4*>(smem + (c0 + k) * 512 + (((r0 >> 2) ^ swz(c0 + k)) << 4));
      if (n0 + c0 < a.N) {   // (N % 4 == 0: a block of four columns is inside or outside as a whole)
        float* cp = C + (long)(m0 + r0) * a.ldc + n0 + c0;
        // Non-temporal (round 6): a whole-tile product streams its C (619 MB for the dcol product) through an L2 whose other
        // tenant is the shared operand every tile re-reads; the product alone 0.469 -> 0.362 ms on a fresh C, the layer's
        // backward 1.47 -> 1.44 (its consumers then find less of dcol in the memory-side cache): profiles/r06h_*
        typedef float f4v __attribute__((ext_vector_type(4)));
        const f4v o0 = {q[0].x, q[1].x, q[2].x, q[3].x}, o1 = {q[0].y, q[1].y, q[2].y, q[3].y};
        const f4v o2 = {q[0].z, q[1].z, q[2].z, q[3].z}, o3 = {q[0].w, q[1].w, q[2].w, q[3].w};
        if (a.nt_store) {
          if (m0 + r0 + 0 < a.M) __builtin_nontemporal_store(o0, reinterpret_cast<f4v*>(cp));
          if (m0 + r0 + 1 < a.M) __builtin_nontemporal_store(o1, reinterpret_cast<f4v*>(cp + a.ldc));
          if (m0 + r0 + 2 < a.M) __builtin_nontemporal_store(o2, reinterpret_cast<f4v*>(cp + 2L * a.ldc));
          if (m0 + r0 + 3 < a.M) __builtin_nontemporal_store(o3, reinterpret_cast<f4v*>(cp + 3L * a.ldc));
        } else {
          if (m0 + r0 + 0 < a.M) *reinterpret_cast<f4v*>(cp) = o0;
          if (m0 + r0 + 1 < a.M) *reinterpret_cast<f4v*>(cp + a.ldc) = o1;
          if (m0 + r0 + 2 < a.M) *reinterpret_cast<f4v*>(cp + 2L * a.ldc) = o2;
          if (m0 + r0 + 3 < a.M) *reinterpret_cast<f4v*>(cp + 3L * a.ldc) = o3;
        }
      }
    }
    return;
  }
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
        }
      }
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
    g.vec_store = (((uintptr_t)g.C & 15) == 0 && g.ldc % 4 == 0 && g.strideC % 4 == 0 && g.N % 4 == 0 &&
                   tuning("deform_gemm_vecstore", 1) == 1) ? 1 : 0;
    g.nt_store = tuning("deform_gemm_nt", 1);
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

AbsSeg absmax_seg(const float* p, long rows, int cols, long ld, long bstride, int batch, unsigned* out) {
  AbsSeg a{p, rows, cols, ld, bstride, batch, out, 0};
  const long n = rows * cols * batch;
  if (n <= 0 || !p) return a;
  long blocks = (n + 256 * 16 - 1) / (256 * 16);   // >= 16 floats per lane
  if (blocks > 4 * kNumCU) blocks = 4 * kNumCU;
  a.blocks = (int)blocks;
  return a;
}

void launch_absmax(AbsSeg s0, AbsSeg s1, AbsSeg s2, hipStream_t st) {
  const int blocks = s0.blocks + s1.blocks + s2.blocks;
  if (blocks <= 0) return;
  hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)blocks), dim3(256), 0, st, s0, s1, s2);
}

}  // namespace sd

using namespace sd;

int sd::gemm_f32_impl(int transA, int transB, int M, int N, int K, const float* A, int lda,
                         long strideA, const float* B, int ldb, long strideB, float* C, int ldc,
                         long strideC, int batch, int accumulate, const unsigned* amax, void* stream) {
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
