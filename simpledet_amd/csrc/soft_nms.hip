// Batched soft-NMS (and bbox_overlaps) for gfx950.
//   reference: operator_py/cython/cpu_nms.pyx:98-203 (soft_nms: in-place selection sort with
//              score decay and swap-with-last removal), operator_py/nms.py:5-16 (wrapper),
//              detection_test.py:233-267 (one call per (image, class), process pool);
//              operator_py/cython/bbox.pyx:31-72 (bbox_overlaps_cython).
// The algorithm is N dependent steps per problem, so throughput comes from running 1000+ problems
// at once, each entirely on-chip: one workgroup per problem, boxes as six SoA planes in LDS
// (24 B/box), and per step only two workgroup barriers:
//   read the arg-max (already reduced during the previous step's rescoring pass) -> barrier ->
//   rescore every remaining box against it, fold the NEXT arg-max into the same pass, ballot the
//   "fell below threshold" flags -> barrier.
// Removal keeps the reference's exact order semantics (hole filled by the last box, which is then
// examined in the hole): because every box's decay depends only on the selected box, all flags are
// computed in parallel and the reference's two-pointer walk is replayed on the 64-bit flag words by
// one wave, moving only the boxes that really move.
// Arithmetic follows the Cython-generated C expression by expression: "+ 1" is a DOUBLE add there
// (the literal becomes 1.0), products of two such terms are double products narrowed on assignment.
#include "common.h"
#include "../../include/simpledet_ops.h"
#include <math.h>

namespace sd {

struct SoftArgs {
  const float* dets;
  const int* counts;
  float* out_dets;
  int* out_inds;
  int* out_counts;
  int P, Nmax;
  float sigma, Nt, thr;
  int method;
};

__device__ __forceinline__ float pmin(float a, float b) { return a <= b ? a : b; }  // cpu_nms.pyx:27
__device__ __forceinline__ float pmax(float a, float b) { return a >= b ? a : b; }  // cpu_nms.pyx:30

// A candidate of the arg-max "maxscore < boxes[pos, 4]" scanning upward from i (the first maximum wins, NaN
// never wins) as ONE 64-bit key: the score as an order-preserving 32-bit integer in the high word, the
// complemented position in the low word; the larger key is the better candidate (higher score, then lower
// position), 0 is "none".  One 64-bit compare + two selects per reduction step instead of the five compares
// of a (score, position) pair -- the kernel is bound by its VALU instruction count (round 5: SQ_ACTIVE_INST_VALU
// 76 % of the SIMD cycles with five problems per CU, profiles/r05g_ops_pmc_summary.json).
//   -0.0 and +0.0 compare equal as floats, so -0.0 enters as +0.0; NaN scores never become candidates.
typedef unsigned long long Cand;
__device__ __forceinline__ Cand cand_of(float s, int pos) {
  const unsigned b = __float_as_uint(s + 0.0f);
  const unsigned ord = b ^ ((unsigned)((int)b >> 31) | 0x80000000u);   // monotone in s over the non-NaN floats, > 0
  return ((Cand)ord << 32) | (unsigned)(0x7fffffff - pos);
}
__device__ __forceinline__ int cand_pos(Cand c) { return c ? 0x7fffffff - (int)(unsigned)c : -1; }
__device__ __forceinline__ float cand_score(Cand c) {
  const unsigned ord = (unsigned)(c >> 32);
  return __uint_as_float(ord ^ ((ord >> 31) ? 0x80000000u : 0xffffffffu));
}
__device__ __forceinline__ Cand better(Cand a, Cand b) { return b > a ? b : a; }
// Wave arg-max on the DPP network (row shifts inside 16-lane rows, then two row broadcasts; the
// result lands in lane 63): VALU latency per step instead of a ds_bpermute round trip per shuffle.
// Returns a wave-uniform value.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ Cand dpp_step(Cand c) {
  // lanes without a source keep "none" (0), which never wins
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)c, CTRL, ROW_MASK, 0xf, false);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(c >> 32), CTRL, ROW_MASK, 0xf, false);
  return better(c, ((Cand)hi << 32) | lo);
}
__device__ __forceinline__ Cand wave_best_dpp(Cand c) {
  c = dpp_step<0x111, 0xf>(c);  // row_shr:1
  c = dpp_step<0x112, 0xf>(c);  // row_shr:2
  c = dpp_step<0x114, 0xf>(c);  // row_shr:4
  c = dpp_step<0x118, 0xf>(c);  // row_shr:8
  c = dpp_step<0x142, 0xa>(c);  // row_bcast:15 into rows 1 and 3
  c = dpp_step<0x143, 0xc>(c);  // row_bcast:31 into rows 2 and 3
  const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)c, 63);
  const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(c >> 32), 63);
  return ((Cand)hi << 32) | lo;
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void soft_nms_kernel(SoftArgs a) {
  constexpr int NW = THREADS / kWave;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int p = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & (kWave - 1);
  const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
  const int Nmax = a.Nmax;
  float* X1 = smem;
  float* Y1 = X1 + Nmax;
  float* X2 = Y1 + Nmax;
  float* Y2 = X2 + Nmax;
  float* S = Y2 + Nmax;
  int* IND = reinterpret_cast<int*>(S + Nmax);
  // after the six planes: flag words, per-wave partial arg-max, control words
  unsigned long long* FW = reinterpret_cast<unsigned long long*>(smem + (((size_t)6 * Nmax + 1) & ~(size_t)1));
  const int nwords = (Nmax + 63) / 64 + 1;
  Cand* PK = reinterpret_cast<Cand*>(FW + nwords);   // per-wave partial arg-max (keys)
  int* CTRL = reinterpret_cast<int*>(PK + NW);  // [0] = N, [1] = step stamp of the last removal
  int* LIST = CTRL + 4;
  const int qcap = (((Nmax + kWave - 1) / kWave + NW - 1) / NW) * kWave;  // queue words per wave

  int n = a.counts ? a.counts[p] : Nmax;
  n = n < 0 ? 0 : (n > Nmax ? Nmax : n);
  const float* src = a.dets + (long)p * Nmax * 5;
  for (int e = tid; e < n * 5; e += THREADS) {
    const int pos = e / 5, k = e % 5;
    smem[k * Nmax + pos] = src[e];
  }
  for (int pos = tid; pos < n; pos += THREADS) IND[pos] = pos;
  if (tid == 0) {
    CTRL[0] = n;
    CTRL[1] = -1;
  }
  __syncthreads();

  int N = n;
  bool have_best = false;
  for (int i = 0; i < n; ++i) {  // range(N) is evaluated once (cpu_nms.pyx:115)
    if (i >= N) break;
    if (!have_best) {
      Cand c = 0;
      for (int pos = i + wave * kWave + lane; pos < N; pos += THREADS) {
        const float sc = S[pos];
        if (sc == sc) c = better(c, cand_of(sc, pos));
      }
      c = wave_best_dpp(c);
      if (lane == 0) PK[wave] = c;
      __syncthreads();
    }
    // ---- selected box: arg-max over [i, N), box i wins ties and is immune to NaN comparisons ----
    Cand best = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) best = better(best, PK[w]);
    const float si = S[i];
    // `si < best score`: box i sits at the lowest position, so its key beats every candidate of an equal score
    const int maxpos = (!(si == si) || best <= cand_of(si, i)) ? i : cand_pos(best);
    // registers: the selected box (t*) and the old box i, which goes to position maxpos
    const float tx1 = X1[maxpos], ty1 = Y1[maxpos], tx2 = X2[maxpos], ty2 = Y2[maxpos];
    const float ts = S[maxpos];
    const int ti = IND[maxpos];
    const float ix1 = X1[i], iy1 = Y1[i], ix2 = X2[i], iy2 = Y2[i];
    const int ii = IND[i];
    __syncthreads();  // everyone holds t* / old i before anybody overwrites them
    if (tid == 0 && maxpos != i) {
      X1[i] = tx1; Y1[i] = ty1; X2[i] = tx2; Y2[i] = ty2; S[i] = ts; IND[i] = ti;
    }
    const double tarea = ((double)(tx2 - tx1) + 1.0) * ((double)(ty2 - ty1) + 1.0);

    // ---- pass 1 (all boxes, cheap): does the box overlap the selected one at all?  In the
    // reference a box is rescored only when iw > 0 and ih > 0; (float)((double)d + 1.0) > 0 is
    // exactly d > -1 for a float d.  Typically ~1% of the boxes overlap, so they are queued and the
    // expensive IoU / decay arithmetic (double adds, IEEE divide) runs once per step on a dense set
    // of lanes instead of in every 64-box chunk.  Untouched boxes enter the next arg-max here.
    Cand c = 0;
    const int M = N - (i + 1);
    // every wave queues the overlapping boxes of ITS chunks in its own segment (length in a
    // register: no LDS atomic, and no barrier between the two passes -- pass 2 of a wave only
    // touches boxes and flag words of that wave's chunks)
    int* QL = LIST + wave * qcap;
    int nq = 0;
    for (int chunk = wave; chunk * kWave < M; chunk += NW) {
      const int rel = chunk * kWave + lane;
      const int pos = i + 1 + rel;
      bool ovl = false;
      if (rel < M) {
        float x1, y1, x2, y2, s;
        if (pos == maxpos) {  // receives the old box i (the swap of cpu_nms.pyx:136-150)
          x1 = ix1; y1 = iy1; x2 = ix2; y2 = iy2; s = si;
          X1[pos] = x1; Y1[pos] = y1; X2[pos] = x2; Y2[pos] = y2; IND[pos] = ii; S[pos] = s;
        } else {
          x1 = X1[pos]; y1 = Y1[pos]; x2 = X2[pos]; y2 = Y2[pos]; s = S[pos];
        }
        const float dw = pmin(tx2, x2) - pmax(tx1, x1), dh = pmin(ty2, y2) - pmax(ty1, y1);
        ovl = dw > -1.f && dh > -1.f;
        if (!ovl && s == s) c = better(c, cand_of(s, pos));
      }
      const unsigned long long bal = __ballot(ovl);
      if (ovl) QL[nq + __popcll(bal & ((1ull << lane) - 1))] = pos;
      nq += __popcll(bal);
      if (lane == 0) FW[chunk] = 0ull;
    }
    // ---- pass 2 (queued boxes only): IoU, decay, removal flag ----
    wave_lds_sync();  // the queue and the old box i were written by other lanes of this wave
    bool any_removed = false;
    for (int e = lane; e < nq; e += kWave) {
      const int pos = QL[e], rel = pos - (i + 1);
      const float x1 = X1[pos], y1 = Y1[pos], x2 = X2[pos], y2 = Y2[pos], s = S[pos];
      const float area = (float)(((double)(x2 - x1) + 1.0) * ((double)(y2 - y1) + 1.0));
      const float iw = (float)((double)(pmin(tx2, x2) - pmax(tx1, x1)) + 1.0);
      const float ih = (float)((double)(pmin(ty2, y2) - pmax(ty1, y1)) + 1.0);
      const float ua = (float)((tarea + (double)area) - (double)(iw * ih));
      const float ov = iw * ih / ua;
      float weight;
      if (a.method == 1) weight = ov > a.Nt ? (float)(1.0 - (double)ov) : 1.f;
      else if (a.method == 2) weight = (float)exp((double)(-(ov * ov) / a.sigma));
      else weight = ov > a.Nt ? 0.f : 1.f;
      const float ns = weight * s;
      S[pos] = ns;
      if (ns < a.thr) {
        any_removed = true;
        atomicOr(&FW[rel >> 6], 1ull << (rel & 63));
      } else if (ns == ns) {
        c = better(c, cand_of(ns, pos));
      }
    }
    c = wave_best_dpp(c);
    if (lane == 0) PK[wave] = c;
    if (any_removed) CTRL[1] = i;
    __syncthreads();
    have_best = true;
    if (CTRL[1] == i) {
      // ---- replay "swap with the last box, shrink, re-examine" on the flag words (one wave) ----
      const int nw = (M + 63) / 64;
      bool keep_best = false;  // uniform over the workgroup (depends on nw only)
      if (nw <= kWave) {
        // Fast path (<= 4096 remaining boxes).  The flag words sit in the lanes of wave 0 and are
        // read with v_readlane, so the hole / last-box walk is scalar code without LDS round
        // trips; it only RECORDS the moves (hole <- last surviving box), lane k holding move k.
        // The moves never depend on each other (every source lies behind every hole), so a batch
        // of up to 64 is then executed by the lanes in parallel.  The arg-max of the remaining
        // boxes folded during this step stays valid: a moved box changes position only, so the
        // best position is re-derived from the moved boxes that carry the best score (ties keep
        // "lowest position wins") instead of rescanning every score.
        keep_best = true;
        if (wave == 0) {
          const unsigned long long fw = lane < nw ? FW[lane] : 0ull;
          auto word = [&](int w) {
            const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)fw, w);
            const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(fw >> 32), w);
            return ((unsigned long long)hi << 32) | lo;
          };
          auto flag = [&](int r) { return (word(r >> 6) >> (r & 63)) & 1ull; };
          auto next_hole = [&](int from) {  // first flagged rel position >= from, or a huge value
            int w = from >> 6;
            if (w >= nw) return 1 << 30;
            unsigned long long m = word(w) & (~0ull << (from & 63));
            while (!m) {
              if (++w >= nw) return 1 << 30;
              m = word(w);
            }
            return (w << 6) + __ffsll((long long)m) - 1;
          };
          Cand gb = 0;  // arg-max of the remaining boxes, positions before the moves
#pragma unroll
          for (int w = 0; w < NW; ++w) gb = better(gb, PK[w]);
          int bpos = cand_pos(gb);
          const float gbs = cand_score(gb);
          int mv_h = 0, mv_s = 0;  // this lane's move: absolute positions hole <- source
          auto flush = [&](int cnt) {
            if (cnt == 0) return;
            int cand = 0x7fffffff;
            bool is_best = false;
            if (lane < cnt) {
              float v[6];
#pragma unroll
              for (int f = 0; f < 6; ++f) v[f] = smem[f * Nmax + mv_s];
#pragma unroll
              for (int f = 0; f < 6; ++f) smem[f * Nmax + mv_h] = v[f];
              is_best = mv_s == bpos;
              if (gb && v[4] == gbs) cand = mv_h;  // carries the best score (never NaN)
            }
            if (__any(cand != 0x7fffffff)) {  // rare: a moved box carries the best score
              if (__any(is_best)) bpos = 0x7fffffff;  // the recorded best itself moved: it is in `cand`
#pragma unroll
              for (int o = 32; o > 0; o >>= 1) cand = iminr(cand, __shfl_xor(cand, o));
              bpos = iminr(bpos, cand);
            }
          };
          int Mc = M, nmv = 0;
          int hole = next_hole(0);
          while (hole < Mc) {
            --Mc;  // the last box leaves its place
            if (Mc == hole) break;       // the hole was the last box
            if (flag(Mc)) continue;      // moved into the hole, examined, removed as well
            if (lane == (nmv & (kWave - 1))) {
              mv_h = i + 1 + hole;
              mv_s = i + 1 + Mc;
            }
            if ((++nmv & (kWave - 1)) == 0) flush(kWave);
            hole = next_hole(hole + 1);
          }
          flush(nmv & (kWave - 1));
          if (lane == 0) {
            CTRL[0] = i + 1 + Mc;
            PK[0] = gb ? ((gb & 0xffffffff00000000ull) | (unsigned)(0x7fffffff - bpos)) : 0;
          }
          if (lane > 0 && lane < NW) PK[lane] = 0;
        }
      } else if (wave == 0) {
        int Mc = M;
        auto flag = [&](int r) { return (FW[r >> 6] >> (r & 63)) & 1ull; };
        auto next_hole = [&](int from) {  // first flagged rel position >= from, or a huge value
          int w = from >> 6;
          if (w >= nw) return 1 << 30;
          unsigned long long m = FW[w] & (~0ull << (from & 63));
          while (!m) {
            if (++w >= nw) return 1 << 30;
            m = FW[w];
          }
          return (w << 6) + __ffsll((long long)m) - 1;
        };
        int hole = next_hole(0);
        while (hole < Mc) {
          --Mc;  // the last box leaves its place
          if (Mc == hole) break;       // the hole was the last box
          if (flag(Mc)) continue;      // moved into the hole, examined, removed as well
          if (lane < 6) smem[lane * Nmax + i + 1 + hole] = smem[lane * Nmax + i + 1 + Mc];
          hole = next_hole(hole + 1);
        }
        if (lane == 0) CTRL[0] = i + 1 + Mc;
      }
      __syncthreads();
      N = CTRL[0];
      have_best = keep_best;  // slow path: positions moved, recompute the arg-max from scratch
    }
  }

  if (tid == 0) a.out_counts[p] = N;
  float* od = a.out_dets + (long)p * Nmax * 5;
  for (int e = tid; e < N * 5; e += THREADS) od[e] = smem[(e % 5) * Nmax + e / 5];
  int* oi = a.out_inds + (long)p * Nmax;
  for (int pos = tid; pos < N; pos += THREADS) oi[pos] = IND[pos];
}

__global__ __launch_bounds__(256) void bbox_overlaps_kernel(const float* boxes, int n,
                                                            const float* query, int k,
                                                            float* overlaps) {
  const long count = (long)n * k;
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < count;
       idx += (long)gridDim.x * blockDim.x) {
    const int kk = (int)(idx % k), nn = (int)(idx / k);
    const float4 b = reinterpret_cast<const float4*>(boxes)[nn];
    const float4 q = reinterpret_cast<const float4*>(query)[kk];
    // Cython builtin min/max on C floats: min(a,b) = b < a ? b : a, max(a,b) = b > a ? b : a
    const float box_area = (float)(((double)(q.z - q.x) + 1.0) * ((double)(q.w - q.y) + 1.0));
    float ov = 0.f;
    const float iw = (float)((double)((q.z < b.z ? q.z : b.z) - (q.x > b.x ? q.x : b.x)) + 1.0);
    if (iw > 0) {
      const float ih = (float)((double)((q.w < b.w ? q.w : b.w) - (q.y > b.y ? q.y : b.y)) + 1.0);
      if (ih > 0) {
        const float ua = (float)(((((double)(b.z - b.x) + 1.0) * ((double)(b.w - b.y) + 1.0)) +
                                  (double)box_area) - (double)(iw * ih));
        ov = iw * ih / ua;
      }
    }
    overlaps[idx] = ov;
  }
}

}  // namespace sd

using namespace sd;

extern "C" int sd_soft_nms_batched(const float* dets, const int32_t* counts, int P, int Nmax,
                                   float sigma, float Nt, float threshold, int method,
                                   float* out_dets, int32_t* out_inds, int32_t* out_counts,
                                   void* stream) {
  SD_REQUIRE(P >= 0 && Nmax >= 0, "negative dimension");
  SD_REQUIRE(method >= 0 && method <= 2, "Unknown soft_nms method: %d", method);
  if (P == 0) return SD_OK;
  SD_REQUIRE(out_counts, "out_counts is null");
  if (Nmax == 0) {
    SD_HIP_CHECK(hipMemsetAsync(out_counts, 0, sizeof(int) * (size_t)P, (hipStream_t)stream));
    return SD_OK;
  }
  SD_REQUIRE(dets && out_dets && out_inds, "null tensor pointer");
  int T = tuning("soft_nms_threads", 256);
  if (T != 64 && T != 128) T = 256;
  const size_t lds = (((size_t)6 * Nmax + 1) & ~(size_t)1) * 4 + ((size_t)(Nmax + 63) / 64 + 1) * 8 +
                     (size_t)(T / kWave) * 8 + 16 +
                     (size_t)((((Nmax + 63) / 64 + T / kWave - 1) / (T / kWave)) * 64) * (T / kWave) * 4;
  SD_REQUIRE(lds <= 160 * 1024, "soft_nms: Nmax=%d needs %zu B of LDS (limit 160 KB)", Nmax, lds);
  SoftArgs a{dets, counts, out_dets, out_inds, out_counts, P, Nmax, sigma, Nt, threshold, method};
#define SD_SOFT(TT)                                                                              \
  do {                                                                                           \
    auto k = soft_nms_kernel<TT>;                                                                \
    if (lds > 64 * 1024)                                                                         \
      SD_HIP_CHECK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                       (int)lds));                                               \
    hipLaunchKernelGGL(k, dim3(P), dim3(TT), lds, (hipStream_t)stream, a);                       \
  } while (0)
  if (T == 64) SD_SOFT(64);
  else if (T == 128) SD_SOFT(128);
  else SD_SOFT(256);
#undef SD_SOFT
  SD_LAUNCH_CHECK();
  return SD_OK;
}

extern "C" int sd_bbox_overlaps(const float* boxes, int n, const float* query_boxes, int k,
                                float* overlaps, void* stream) {
  SD_REQUIRE(n >= 0 && k >= 0, "negative dimension");
  const long count = (long)n * k;
  if (count == 0) return SD_OK;
  SD_REQUIRE(boxes && query_boxes && overlaps, "null tensor pointer");
  SD_REQUIRE((((uintptr_t)boxes | (uintptr_t)query_boxes) & 15) == 0, "boxes must be 16-B aligned");
  const int grid = (int)((count + 255) / 256 < kNumCU * 16 ? (count + 255) / 256 : kNumCU * 16);
  hipLaunchKernelGGL(bbox_overlaps_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, boxes, n,
                     query_boxes, k, overlaps);
  SD_LAUNCH_CHECK();
  return SD_OK;
}
