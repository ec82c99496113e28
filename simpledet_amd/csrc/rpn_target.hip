// RPN anchor-target assignment for gfx950, device resident.
//
// The reference computes these targets in its data loader, per image, in numpy on the host
// (core/detection_input.py AnchorTarget2D :345-565, models/FPN/input.py PyramidAnchorTarget2D
// :9-146, IoU operator_py/cython/bbox.pyx:31-72, box encoding operator_py/bbox_transform.py:52-77)
// and it is the loader bottleneck it works around with a TVM op elsewhere.  Here the whole
// assignment runs on the GPU and reproduces the reference bit for bit, INCLUDING the random
// subsampling: np.random.choice(inds, size, replace=False) of the legacy RandomState is
// inds[permutation(n)[:size]], permutation = Fisher-Yates from the top over MT19937 with masked
// rejection sampling, and the generator state lives in device memory and is advanced exactly as
// numpy would advance it (624 key words + position).
//
// Kernels (one stream, no host involvement):
//   rpn_overlap   one thread per anchor: the anchor box from (level, y, x, a), the inside-image
//                 test, IoU against the valid gt boxes held in LDS with the Cython arithmetic
//                 (float ops, the literal "+ 1" as a double add), first-maximum arg-max; per-gt
//                 maximum over the valid anchors through LDS then global atomic max on the bits
//   rpn_label     label -1/0/1 per anchor (:462-477, including the reference's own looseness: every
//                 anchor whose overlap EQUALS a gt's maximum and is >= min_pos_thr is positive --
//                 with min_pos_thr = 0 a gt box nothing overlaps makes every valid anchor positive),
//                 per-block fg / bg counts; rpn_scan + rpn_lists turn them into the ordered index
//                 lists np.where would return
//   rpn_sample    ONE workgroup, images in order (the generator state carries from image to image); round 6: five
//                 waves that meet through LDS counters only --
//                 * four PRODUCER waves make MT19937's raw sequence (X[j] = X[j-227] ^ tw(X[j-624], X[j-623]):
//                   227 consecutive words are independent, and two steps fit between hand-shakes) into an LDS ring,
//                   raw and tempered, as far ahead as the ring allows
//                 * the CONSUMER wave takes up to 1024 draws per dependent step: candidates (v & mask <= i) by
//                   ballot, sure accepts by a rank bound, the handful of unsettled ones in order with their exact
//                   rank; batch by batch (ballot / popcount fixed point, exact) where the rejection mask is small,
//                   near a mask segment's end and inside the recorded swaps
//                 * only the last `keep` positions of the permutation survive, and they are final
//                   after the first `keep` swaps: those swaps are replayed on a sparse array
//                 so ~330 k dependent draws per image cost ~700 wave steps (2.7 -> 1.4 ms for two images; the rest is
//                 the single consumer wave's ~12 clocks per instruction)
//   rpn_encode    final label, box deltas in double (nonlinear_transform on float64 anchors),
//                 weights; written either in flat all-anchor order or in the loader's final
//                 (A, sum h*w) / (4A, sum h*w) layout
#include "common.h"
#include "../../include/simpledet_ops.h"
#include <math.h>

namespace sd {

constexpr int kRpnMaxLvl = SD_MAX_FPN_LEVELS;
constexpr int kRpnMaxA = 16;  // (kernel arguments are limited to 4 KB: base anchors travel in them)

struct RpnArgs {
  sd_rpn_target_param p;
  float base[kRpnMaxLvl][kRpnMaxA][4];  // base anchors per level (exactly representable)
  int num_fg;                           // int(pos_fraction * image_anchor), evaluated in double
  int A;                                // anchors per location
  int N;                                // anchors per image
  int sumHW;
  const float* im_info;  // (B,3)
  const float* gt;       // (B,M,G)
  int B, M, G;
  int* mt;               // 625 words
  float* cls;
  float* tgt;
  float* wgt;
  int layout;
  // workspace
  float* maxov;          // (B,N)
  int* argmax;           // (B,N)
  signed char* label;    // (B,N)
  unsigned char* keep;   // (B,N) survivors of the subsampling
  unsigned* gtmax;       // (B,M) float bits
  int* blk;              // (B, nblk, 2) fg / bg counts, then exclusive offsets
  int* fg_list;          // (B,N)
  int* bg_list;          // (B,N)
  int* counts;           // (B,4): n_fg, n_bg, fg_sampled, bg_sampled
  int nblk;
};

struct AnchorRef {
  float4 box;
  int lvl, y, x, a, fh, fw, hwoff;
};

// anchor n of an image whose orientation is `portrait` (h >= w: fh = long, fw = short)
__device__ __forceinline__ AnchorRef rpn_anchor(const RpnArgs& a, int n, bool portrait) {
  AnchorRef r;
  int off = 0, hwoff = 0, l = 0, fh = 0, fw = 0;
  for (; l < a.p.nlvl; ++l) {
    fh = portrait ? a.p.long_side[l] : a.p.short_side[l];
    fw = portrait ? a.p.short_side[l] : a.p.long_side[l];
    const int cnt = fh * fw * a.A;
    if (n < off + cnt) break;
    off += cnt;
    hwoff += fh * fw;
  }
  const int idx = n - off;
  r.lvl = l; r.fh = fh; r.fw = fw; r.hwoff = hwoff;
  r.a = idx % a.A;
  const int cell = idx / a.A;
  r.x = cell % fw;
  r.y = cell / fw;
  const float sx = (float)r.x * (float)a.p.stride[l], sy = (float)r.y * (float)a.p.stride[l];
  const float* b = a.base[l][r.a];
  r.box = make_float4(sx + b[0], sy + b[1], sx + b[2], sy + b[3]);
  return r;
}

// operator_py/cython/bbox.pyx:56-72 -- C float arithmetic, the integer literal 1 added as a double
__device__ __forceinline__ float rpn_iou(const float4 b, const float4 q) {
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
  return ov;
}

// valid gt rows (gt[:,0] != -1), order preserved, into LDS; returns their number (all threads)
template <int THREADS>
__device__ int rpn_load_gt(const RpnArgs& a, int img, float4* gbox, int* wsum) {
  const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
  int n_gt = 0;
  for (int base = 0; base < a.M; base += THREADS) {
    const int j = base + tid;
    float4 g = make_float4(-1.f, 0.f, 0.f, 0.f);
    if (j < a.M) {
      const float* p = a.gt + ((long)img * a.M + j) * a.G;
      g = make_float4(p[0], p[1], p[2], p[3]);
    }
    const bool ok = j < a.M && g.x != -1.f;
    const unsigned long long m = __ballot(ok);
    __syncthreads();
    if (lane == 0) wsum[wave] = __popcll(m);
    __syncthreads();
    int off = n_gt, tot = 0;
    for (int w = 0; w < THREADS / kWave; ++w) {
      if (w < wave) off += wsum[w];
      tot += wsum[w];
    }
    if (ok) gbox[off + __popcll(m & ((1ull << lane) - 1))] = g;
    n_gt += tot;
  }
  __syncthreads();
  return n_gt;
}

__device__ __forceinline__ bool rpn_valid(const RpnArgs& a, const float4 b, float h, float w) {
  const float ab = (float)a.p.allowed_border;
  return b.x >= -ab && b.y >= -ab && b.z < w + ab && b.w < h + ab;
}

constexpr int kRpnT = 256;

__global__ __launch_bounds__(kRpnT) void rpn_overlap_kernel(RpnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float4 gbox[];
  __shared__ int wsum[kRpnT / kWave];
  unsigned* gmax = reinterpret_cast<unsigned*>(gbox + a.M);
  const int img = blockIdx.y, tid = threadIdx.x;
  const int n = blockIdx.x * kRpnT + tid;
  const int n_gt = rpn_load_gt<kRpnT>(a, img, gbox, wsum);
  for (int j = tid; j < n_gt; j += kRpnT) gmax[j] = 0u;
  __syncthreads();
  const float h = a.im_info[img * 3], w = a.im_info[img * 3 + 1];
  if (n < a.N) {
    const AnchorRef r = rpn_anchor(a, n, h >= w);
    float maxv = 0.f;
    int maxi = 0;
    if (rpn_valid(a, r.box, h, w)) {
      for (int j = 0; j < n_gt; ++j) {
        const float o = rpn_iou(r.box, gbox[j]);
        if (j == 0 || o > maxv) {  // np.argmax: first maximum
          maxv = o;
          maxi = j;
        }
        if (o > 0.f) atomicMax(&gmax[j], __float_as_uint(o));
      }
    }
    a.maxov[(long)img * a.N + n] = maxv;
    a.argmax[(long)img * a.N + n] = maxi;
    a.keep[(long)img * a.N + n] = 0;
  }
  __syncthreads();
  for (int j = tid; j < n_gt; j += kRpnT)
    if (gmax[j]) atomicMax(&a.gtmax[(long)img * a.M + j], gmax[j]);
}

__global__ __launch_bounds__(kRpnT) void rpn_label_kernel(RpnArgs a) {
  extern __shared__ __attribute__((aligned(16))) float4 gbox[];
  __shared__ int wsum[kRpnT / kWave];
  __shared__ int cnt[2];
  const int img = blockIdx.y, tid = threadIdx.x;
  const int n = blockIdx.x * kRpnT + tid;
  const int n_gt = rpn_load_gt<kRpnT>(a, img, gbox, wsum);
  if (tid < 2) cnt[tid] = 0;
  __syncthreads();
  const float h = a.im_info[img * 3], w = a.im_info[img * 3 + 1];
  int lab = -1;
  if (n < a.N) {
    const AnchorRef r = rpn_anchor(a, n, h >= w);
    if (rpn_valid(a, r.box, h, w)) {
      if (n_gt > 0) {
        const float mo = a.maxov[(long)img * a.N + n];
        if (mo < a.p.neg_thr) lab = 0;
        bool is_gt_arg = false;
        for (int j = 0; j < n_gt; ++j) {
          const float o = rpn_iou(r.box, gbox[j]);
          const float gm = __uint_as_float(a.gtmax[(long)img * a.M + j]);
          if (o == gm && o >= a.p.min_pos_thr) is_gt_arg = true;
        }
        if (is_gt_arg) lab = 1;
        if (mo >= a.p.pos_thr) lab = 1;
      } else {
        lab = 0;
      }
    }
    a.label[(long)img * a.N + n] = (signed char)lab;
  }
  const unsigned long long mf = __ballot(lab == 1), mb = __ballot(lab == 0);
  if ((tid & (kWave - 1)) == 0) {
    atomicAdd(&cnt[0], __popcll(mf));
    atomicAdd(&cnt[1], __popcll(mb));
  }
  __syncthreads();
  if (tid < 2) a.blk[((long)img * a.nblk + blockIdx.x) * 2 + tid] = cnt[tid];
}

// exclusive scan of the per-block counts (one workgroup per image); totals into counts
__global__ __launch_bounds__(1024) void rpn_scan_kernel(RpnArgs a) {
  __shared__ int wtot[2][16];
  const int img = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int run[2] = {0, 0};
  for (int base = 0; base < a.nblk; base += 1024) {
    const int b = base + tid;
    int v[2] = {0, 0};
    if (b < a.nblk) {
      v[0] = a.blk[((long)img * a.nblk + b) * 2];
      v[1] = a.blk[((long)img * a.nblk + b) * 2 + 1];
    }
    int incl[2] = {v[0], v[1]};
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t0 = __shfl_up(incl[0], o), t1 = __shfl_up(incl[1], o);
      if (lane >= o) { incl[0] += t0; incl[1] += t1; }
    }
    __syncthreads();
    if (lane == 63) { wtot[0][wave] = incl[0]; wtot[1][wave] = incl[1]; }
    __syncthreads();
    int off[2] = {run[0], run[1]}, tot[2] = {0, 0};
    for (int wv = 0; wv < 16; ++wv) {
      if (wv < wave) { off[0] += wtot[0][wv]; off[1] += wtot[1][wv]; }
      tot[0] += wtot[0][wv]; tot[1] += wtot[1][wv];
    }
    if (b < a.nblk) {
      a.blk[((long)img * a.nblk + b) * 2] = off[0] + incl[0] - v[0];
      a.blk[((long)img * a.nblk + b) * 2 + 1] = off[1] + incl[1] - v[1];
    }
    run[0] += tot[0]; run[1] += tot[1];
  }
  if (tid == 0) {
    a.counts[img * 4] = run[0];
    a.counts[img * 4 + 1] = run[1];
    a.counts[img * 4 + 2] = 0;
    a.counts[img * 4 + 3] = 0;
  }
}

__global__ __launch_bounds__(kRpnT) void rpn_lists_kernel(RpnArgs a) {
  __shared__ int wcnt[2][kRpnT / kWave];
  const int img = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n = blockIdx.x * kRpnT + tid;
  const int lab = n < a.N ? (int)a.label[(long)img * a.N + n] : -1;
  const unsigned long long mf = __ballot(lab == 1), mb = __ballot(lab == 0);
  if (lane == 0) { wcnt[0][wave] = __popcll(mf); wcnt[1][wave] = __popcll(mb); }
  __syncthreads();
  int off0 = a.blk[((long)img * a.nblk + blockIdx.x) * 2];
  int off1 = a.blk[((long)img * a.nblk + blockIdx.x) * 2 + 1];
  for (int wv = 0; wv < wave; ++wv) { off0 += wcnt[0][wv]; off1 += wcnt[1][wv]; }
  const unsigned long long lt = (1ull << lane) - 1;
  if (lab == 1) a.fg_list[(long)img * a.N + off0 + __popcll(mf & lt)] = n;
  if (lab == 0) a.bg_list[(long)img * a.N + off1 + __popcll(mb & lt)] = n;
}

// ---- MT19937 (numpy legacy RandomState): a producer wave and a consumer wave -----------------------
// Round 6.  Until round 5 ONE wave twisted the state and consumed it (2.7 ms for two images: a lone wave pays
// ~12 clocks per dependent instruction, and twist and rejection chain were one dependency chain).  The generator's
// raw output is a pure sequence -- X[j] = X[j - 227] ^ tw(X[j - 624], X[j - 623]) for j >= 624, X[0 .. 623] = the
// state's key words; all three sources lie >= 227 words back, so 227 consecutive words are independent -- and only
// WHERE in it each draw is taken depends on the data.  So the other waves of the workgroup produce the sequence
// into an LDS ring, 227 words per step, as far ahead as the ring allows, and the first wave consumes it (tempering
// on read); four waves share the production, 57 words of a step each.  The two meet through LDS counters only (release / acquire at workgroup scope; no s_barrier: the waves
// of one workgroup are co-resident by construction, the same guarantee a barrier rests on).  Every wait is a bounded
// spin: on overrun the kernel stops and raises `counts[B * 4]` (the host entry point does not read it back --
// a wrong sample shows in the tests, a hang would cost a GPU).
constexpr int kMtRing = 4096;        // words (16 KB raw + 16 KB tempered); the producers stay < kMtRing ahead of what the consumer still needs
constexpr int kMtSpinCap = 1 << 22;  // ~seconds of s_sleep polling
constexpr int kMtProd = 4;           // producer waves: wave w makes elements [57 w, 57 w + 57) of every 227-word step
constexpr int kMtSlice = 57;         // (227 = 3 * 57 + 56)

struct MtStream {
  unsigned* ring;   // LDS: word j of the sequence at ring[j & (kMtRing - 1)] (raw: what the recurrence and the state need)
  unsigned* tring;  // LDS: the same word tempered (what a draw returns), written by the producers beside it
  int* ctl;         // LDS: [0..3] steps completed by the producer waves, [4] first word the consumer still needs,
                    //      [5] stop, [6] overrun
  int pos;          // consumer: index of the next output (wave uniform)
};

__device__ __forceinline__ int mt_ld(const int* p) {
  return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void mt_st(int* p, int v) {
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// steps every producer wave has completed (one 16-byte LDS read)
__device__ __forceinline__ int mt_steps_done(const int* ctl) {
  typedef int v4i __attribute__((ext_vector_type(4)));
  const v4i d = *reinterpret_cast<const volatile v4i*>(ctl);
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
  return iminr(iminr(d.x, d.y), iminr(d.z, d.w));
}

__device__ __forceinline__ unsigned mt_temper(unsigned y) {
  y ^= y >> 11;
  y ^= (y << 7) & 0x9d2c5680u;
  y ^= (y << 15) & 0xefc60000u;
  y ^= y >> 18;
  return y;
}

// Producer wave w: elements [57 w, 57 w + 57) of every step, TWO steps per hand-shake.  Word j = 624 + 227 s + e
// reads j - 227 (the same element of step s - 1: this wave's own word -- inside a pair it never leaves the
// register) and j - 624 / j - 623 = elements e + 57 / e + 58 of step s - 3 (or e - 170 / e - 169 of step s - 2):
// other waves' words, at least two steps back.  So once every wave has completed step s - 1, steps s AND s + 1 can
// be made without another look at the others: one round of flag reads per 454 words.  (A hand-shake is two or three
// dependent LDS round trips of a wave that issues an instruction every ~12 clocks: with one per step the four
// producers delivered 0.23 words per clock against the 0.45 the consumer takes.)  Runs until ctl[5] is raised.
__device__ void mt_produce(unsigned* ring, unsigned* tring, int* ctl, int w, int lane) {
  typedef int v4i __attribute__((ext_vector_type(4)));
  const int e = w * kMtSlice + lane;
  const bool mine = lane < kMtSlice && e < 227;
  auto tw = [](unsigned far, unsigned a, unsigned b) {
    const unsigned y = (a & 0x80000000u) | (b & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  };
  for (int s = 0;; s += 2) {
    const int made = 624 + 227 * s;   // first word of the pair of steps
    int spins = 0;
    while (true) {
      const v4i d = *reinterpret_cast<const volatile v4i*>(ctl);       // steps completed by the four producers
      const v4i c = *reinterpret_cast<const volatile v4i*>(ctl + 4);   // consumer floor, stop, overrun
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      if (c.y) return;
      // word j overwrites word j - kMtRing: everything from the consumer's floor on stays
      if (made + 454 <= c.x + kMtRing - 64 && iminr(iminr(d.x, d.y), iminr(d.z, d.w)) >= s) break;
      __builtin_amdgcn_s_sleep(1);
      if (++spins > kMtSpinCap) { mt_st(ctl + 6, 1); return; }
    }
    if (mine) {
      const int j = made + e, j1 = j + 227;
      const unsigned f0 = ring[(j - 227) & (kMtRing - 1)];
      const unsigned a0 = ring[(j - 624) & (kMtRing - 1)], b0 = ring[(j - 623) & (kMtRing - 1)];
      const unsigned a1 = ring[(j1 - 624) & (kMtRing - 1)], b1 = ring[(j1 - 623) & (kMtRing - 1)];
      const unsigned x0 = tw(f0, a0, b0), x1 = tw(x0, a1, b1);
      ring[j & (kMtRing - 1)] = x0;
      ring[j1 & (kMtRing - 1)] = x1;
      // (tempered beside it: the consumer is a lone wave too, every instruction taken off it counts)
      tring[j & (kMtRing - 1)] = mt_temper(x0);
      tring[j1 & (kMtRing - 1)] = mt_temper(x1);
    }
    mt_st(ctl + w, s + 2);   // (release: behind the ring writes)
  }
}

// consumer: words [pos, pos + n) are there (false on overrun)
__device__ __forceinline__ bool mt_need(MtStream& m, int n) {
  int spins = 0;
  while (624 + 227 * mt_steps_done(m.ctl) < m.pos + n) {
    __builtin_amdgcn_s_sleep(1);
    if (++spins > kMtSpinCap || mt_ld(m.ctl + 6)) { mt_st(m.ctl + 6, 1); return false; }
  }
  return true;
}

constexpr int kRpnMaxKeep = 1024;

// NB batches of 64 draws in ONE dependent step, exactly.  Draw t accepts iff v_t <= i - A(t), A(t) = accepts in
// front of it.  The CANDIDATES (v <= i) are the only draws that can accept, and the number of candidates up to the
// end of a draw's batch bounds its A(t) from above: v <= i - that is a sure accept.  What is left -- candidates
// with i - bound < v <= i, about bound / mask of them -- is a handful per trip; they are settled one after the
// other in order with their exact rank (each rejection lowers the A of everything behind it by one).  Per batch
// that is one LDS read, two compares and a few scalar operations, independent of the other batches.  Returns the
// number of accepts, or -1 when the trip would leave the mask's segment (the caller then goes batch by batch).
template <int NB>
__device__ __forceinline__ int mt_trip(const MtStream& m, int i, int lo, unsigned mask, int lane) {
  int vv[NB];
#pragma unroll
  for (int j = 0; j < NB; ++j) vv[j] = (int)(m.tring[(m.pos + j * kWave + lane) & (kMtRing - 1)] & mask);
  unsigned long long cand[NB], amb[NB];
  int base[NB + 1];
  base[0] = 0;
  unsigned long long anyamb = 0;
#pragma unroll
  for (int j = 0; j < NB; ++j) {
    cand[j] = __ballot(vv[j] <= i);
    base[j + 1] = base[j] + __popcll(cand[j]);
    amb[j] = __ballot(vv[j] > i - base[j + 1]) & cand[j];
    anyamb |= amb[j];
  }
  int K = base[NB];
  if (anyamb) {
    int rej = 0;
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      unsigned long long todo = amb[j];
      while (todo) {
        const int t = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int vt = __builtin_amdgcn_readlane(vv[j], t);
        const int rt = base[j] + __popcll(cand[j] & ((1ull << t) - 1));
        if (vt > i - rt + rej) ++rej;
      }
    }
    K -= rej;
  }
  return K < i - lo + 1 ? K : -1;
}

// permutation(n) of the legacy RandomState, of which only the last `keep` entries are wanted
// (choice(inds, n - keep, replace=False) disables the FIRST n - keep): fills surv[0..keep) with the
// list positions that survive and advances the generator by exactly the draws numpy consumes.
__device__ void rpn_sample(MtStream& m, int n, int keep, int* jrec, int* hp, int* hv, int* surv, int lane) {
  int i = n - 1;
  const unsigned long long lt = (1ull << lane) - 1;
  while (i >= 1) {
    // the consumer no longer needs anything before the block its last draw came from (the final state is that block)
    mt_st(m.ctl + 4, m.pos > 624 ? m.pos - 624 : 0);
    // Trip size by mask: the unsettled candidates of an L-draw trip number ~L * (L / 2) / mask -- a handful for 1024
    // draws from mask 2^17 on (where four fifths of the draws are spent), for 512 from 2^15, for 128 from 2^11;
    // below that, and inside the recorded swaps, batch by batch.
    const bool past = (n - 1) - i >= keep;
    const int nb = !past || i < 1024 ? 1 : i < 16384 ? 2 : i < 65536 ? 8 : 16;
    if (!mt_need(m, nb * kWave)) return;
    unsigned mask = (unsigned)i;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    const int lo = (int)(mask >> 1) + 1;  // steps i in [lo, mask] share this rejection mask
    if (nb > 1) {
      const int K = nb == 16 ? mt_trip<16>(m, i, lo, mask, lane) : nb == 8 ? mt_trip<8>(m, i, lo, mask, lane)
                                                                          : mt_trip<2>(m, i, lo, mask, lane);
      if (K >= 0) {
        i -= K;
        m.pos += nb * kWave;
        continue;
      }
    }
    const unsigned v = m.tring[(m.pos + lane) & (kMtRing - 1)] & mask;
    bool acc = v <= (unsigned)i;
    unsigned long long bits = __ballot(acc);
    while (true) {  // ballot / popcount fixed point: exact, 1-2 rounds
      const int c = __popcll(bits & lt);
      const bool acc2 = (long)v <= (long)i - c;
      const unsigned long long b2 = __ballot(acc2);
      acc = acc2;
      if (b2 == bits) break;
      bits = b2;
    }
    int K = __popcll(bits);
    const int R = i - lo + 1;  // accepts this mask still serves
    int consumed = kWave;
    if (K >= R) {
      // the R-th accept ends the segment; later draws of the batch are re-read with the next mask
      // (also when it is the batch's last accept: the rejected draws behind it were judged with
      // the old mask, numpy judges them with the halved one -- found by the generator fuzz test)
      unsigned long long b = bits;
      for (int s = 1; s < R; ++s) b &= b - 1;
      const int last = __ffsll((long long)b) - 1;
      consumed = last + 1;
      bits &= (2ull << last) - 1ull;
      acc = acc && lane <= last;
      K = R;
    }
    if (acc) {
      const int step = (n - 1) - (i - __popcll(bits & lt));  // 0-based index of this swap
      if (step < keep) jrec[step] = (int)v;
    }
    i -= K;
    m.pos += consumed;
  }
  wave_lds_sync();
  // replay the first `keep` swaps on a sparse array: position p holds p unless a swap wrote it
  for (int s = 0; s < keep; ++s) {
    const int is = n - 1 - s, js = jrec[s];
    int vi = is, vj = js, fi = -1, fj = -1;
    for (int base = ((s - 1) / kWave) * kWave; base >= 0 && s > 0; base -= kWave) {
      const int e = base + lane;
      const int hpe = e < s ? hp[e] : -2;
      if (fi < 0) {
        const unsigned long long mi = __ballot(hpe == is);
        if (mi) fi = base + 63 - __clzll((long long)mi);
      }
      if (fj < 0) {
        const unsigned long long mj = __ballot(hpe == js);
        if (mj) fj = base + 63 - __clzll((long long)mj);
      }
      if (fi >= 0 && fj >= 0) break;
    }
    if (fi >= 0) vi = hv[fi];
    if (fj >= 0) vj = hv[fj];
    if (lane == 0) {
      surv[s] = vj;  // position is is final: it holds what position js held
      hp[s] = js;    // ... and js now holds what is held
      hv[s] = vi;
    }
    wave_lds_sync();
  }
}

__global__ __launch_bounds__((1 + kMtProd) * kWave) void rpn_sample_kernel(RpnArgs a) {
  __shared__ unsigned ring[kMtRing], tring[kMtRing];
  __shared__ __attribute__((aligned(16))) int ctl[8];
  __shared__ int jrec[kRpnMaxKeep], hp[kRpnMaxKeep], hv[kRpnMaxKeep], surv[kRpnMaxKeep];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  // the key words are the first 624 words of the sequence; a stored position of 624 ("twist before the next draw")
  // is simply the next word
  for (int k = threadIdx.x; k < 624; k += (1 + kMtProd) * kWave) {
    ring[k] = (unsigned)a.mt[k];
    tring[k] = mt_temper(ring[k]);
  }
  if (threadIdx.x < 8) ctl[threadIdx.x] = 0;
  __syncthreads();   // (the only barrier: before the waves part)
  if (wave >= 1) {
    mt_produce(ring, tring, ctl, wave - 1, lane);
    return;
  }
  MtStream m{ring, tring, ctl, a.mt[624]};
  const int pos0 = m.pos;
  const int num = a.p.image_anchor;
  for (int img = 0; img < a.B && !mt_ld(ctl + 6); ++img) {
    int* c = a.counts + img * 4;
    const int n_fg = c[0], n_bg = c[1];
    int fg_left = n_fg;
    if (n_fg > a.num_fg) {
      rpn_sample(m, n_fg, a.num_fg, jrec, hp, hv, surv, lane);
      for (int s = lane; s < a.num_fg; s += kWave)
        a.keep[(long)img * a.N + a.fg_list[(long)img * a.N + surv[s]]] = 1;
      if (lane == 0) c[2] = 1;
      fg_left = a.num_fg;
    }
    const int num_bg = num - fg_left;
    if (n_bg > num_bg) {
      if (num_bg > 0) {
        rpn_sample(m, n_bg, num_bg, jrec, hp, hv, surv, lane);
        for (int s = lane; s < num_bg; s += kWave)
          a.keep[(long)img * a.N + a.bg_list[(long)img * a.N + surv[s]]] = 1;
      } else {
        rpn_sample(m, n_bg, 0, jrec, hp, hv, surv, lane);  // every bg is disabled; draws still consumed
      }
      if (lane == 0) c[3] = 1;
    }
    wave_lds_sync();
  }
  // The state numpy would hold now: the block the last draw came from and the position behind it (624 = "twist
  // before the next draw"); untouched when nothing was drawn.  The block is still in the ring: the producer never
  // overwrites anything from ctl[4] = pos - 624 on.
  if (m.pos != pos0 && !mt_ld(ctl + 6)) {
    const int b0 = ((m.pos - 1) / 624) * 624;
    int spins = 0;   // (the block's tail may still be on its way)
    while (624 + 227 * mt_steps_done(ctl) < b0 + 624 && ++spins <= kMtSpinCap) __builtin_amdgcn_s_sleep(1);
    for (int k = lane; k < 624; k += kWave) a.mt[k] = (int)ring[(b0 + k) & (kMtRing - 1)];
    if (lane == 0) a.mt[624] = m.pos - b0;
  }
  if (lane == 0 && mt_ld(ctl + 6)) a.counts[a.B * 4] = 1;   // overrun: the sample is not valid
  mt_st(ctl + 5, 1);   // the producers may leave
}

__global__ __launch_bounds__(kRpnT) void rpn_encode_kernel(RpnArgs a) {
  const int img = blockIdx.y;
  const int n = blockIdx.x * kRpnT + threadIdx.x;
  if (n >= a.N) return;
  const float h = a.im_info[img * 3], w = a.im_info[img * 3 + 1];
  const AnchorRef r = rpn_anchor(a, n, h >= w);
  const int* c = a.counts + img * 4;
  int lab = (int)a.label[(long)img * a.N + n];
  const bool kept = a.keep[(long)img * a.N + n] != 0;
  if (lab == 1 && c[2] && !kept) lab = -1;
  if (lab == 0 && c[3] && !kept) lab = -1;
  float t[4] = {0.f, 0.f, 0.f, 0.f};
  const float wv = lab == 1 ? 1.f : 0.f;
  if (lab == 1) {
    // nonlinear_transform (operator_py/bbox_transform.py:52-77) on float64 anchors / float32 gt
    const int g = a.argmax[(long)img * a.N + n];
    // the g-th VALID gt row
    int seen = -1;
    const float* gp = nullptr;
    for (int j = 0; j < a.M; ++j) {
      const float* p = a.gt + ((long)img * a.M + j) * a.G;
      if (p[0] != -1.f && ++seen == g) { gp = p; break; }
    }
    const double ex0 = r.box.x, ey0 = r.box.y, ex1 = r.box.z, ey1 = r.box.w;
    const double ew = ex1 - ex0 + 1.0, eh = ey1 - ey0 + 1.0;
    const double ecx = ex0 + 0.5 * (ew - 1.0), ecy = ey0 + 0.5 * (eh - 1.0);
    // gt is float32: gt[:,2] - gt[:,0] is a float32 subtraction, "+ 1.0" promotes per numpy 2 rules
    // (python float is weak: stays float32)
    const float gwf = gp[2] - gp[0] + 1.0f, ghf = gp[3] - gp[1] + 1.0f;
    const float gcxf = gp[0] + 0.5f * (gwf - 1.0f), gcyf = gp[1] + 0.5f * (ghf - 1.0f);
    t[0] = (float)(((double)gcxf - ecx) / (ew + 1e-14));
    t[1] = (float)(((double)gcyf - ecy) / (eh + 1e-14));
    t[2] = (float)log((double)gwf / ew);
    t[3] = (float)log((double)ghf / eh);
  }
  if (a.layout == 0) {
    a.cls[(long)img * a.N + n] = (float)lab;
    float4* tp = reinterpret_cast<float4*>(a.tgt) + (long)img * a.N + n;
    float4* wp = reinterpret_cast<float4*>(a.wgt) + (long)img * a.N + n;
    *tp = make_float4(t[0], t[1], t[2], t[3]);
    *wp = make_float4(wv, wv, wv, wv);
  } else {
    const int col = r.hwoff + r.y * r.fw + r.x;
    a.cls[(long)img * a.N + (long)r.a * a.sumHW + col] = (float)lab;
    for (int k = 0; k < 4; ++k) {
      const long o = ((long)img * a.A * 4 + r.a * 4 + k) * a.sumHW + col;
      a.tgt[o] = t[k];
      a.wgt[o] = wv;
    }
  }
}

static size_t rpn_layout(int B, int N, int M, int nblk, RpnArgs* a, char* base) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off = (off + bytes + 255) / 256 * 256;
    return o;
  };
  const size_t o_mo = take((size_t)B * N * 4), o_am = take((size_t)B * N * 4), o_lb = take((size_t)B * N);
  const size_t o_kp = take((size_t)B * N), o_gm = take((size_t)B * (M > 0 ? M : 1) * 4);
  const size_t o_bk = take((size_t)B * nblk * 2 * 4), o_fg = take((size_t)B * N * 4);
  const size_t o_bg = take((size_t)B * N * 4), o_ct = take((size_t)(B * 4 + 1) * 4);
  if (a) {
    a->maxov = reinterpret_cast<float*>(base + o_mo);
    a->argmax = reinterpret_cast<int*>(base + o_am);
    a->label = reinterpret_cast<signed char*>(base + o_lb);
    a->keep = reinterpret_cast<unsigned char*>(base + o_kp);
    a->gtmax = reinterpret_cast<unsigned*>(base + o_gm);
    a->blk = reinterpret_cast<int*>(base + o_bk);
    a->fg_list = reinterpret_cast<int*>(base + o_fg);
    a->bg_list = reinterpret_cast<int*>(base + o_bg);
    a->counts = reinterpret_cast<int*>(base + o_ct);
  }
  return off;
}

static int rpn_count(const sd_rpn_target_param& p, int* A, int* N, int* sumHW) {
  SD_REQUIRE(p.nlvl >= 1 && p.nlvl <= kRpnMaxLvl, "nlvl=%d outside [1,%d]", p.nlvl, kRpnMaxLvl);
  SD_REQUIRE(p.n_scales >= 1 && p.n_aspects >= 1 && p.n_scales * p.n_aspects <= kRpnMaxA,
             "scales x aspects must be in [1,%d]", kRpnMaxA);
  *A = p.n_scales * p.n_aspects;
  long n = 0, hw = 0;
  for (int l = 0; l < p.nlvl; ++l) {
    SD_REQUIRE(p.stride[l] > 0 && p.short_side[l] > 0 && p.long_side[l] > 0, "level %d: bad size", l);
    hw += (long)p.short_side[l] * p.long_side[l];
  }
  n = hw * *A;
  SD_REQUIRE(n < (1L << 28), "too many anchors");
  *N = (int)n;
  *sumHW = (int)hw;
  return SD_OK;
}

}  // namespace sd

using namespace sd;

extern "C" int sd_rpn_target_num_anchors(const sd_rpn_target_param* p) {
  int A, N, S;
  if (!p || rpn_count(*p, &A, &N, &S)) return -1;
  return N;
}

extern "C" size_t sd_rpn_target_workspace_bytes(const sd_rpn_target_param* p, int B, int M) {
  int A, N, S;
  if (!p || B <= 0 || rpn_count(*p, &A, &N, &S)) return 256;
  return rpn_layout(B, N, M, cdiv(N, kRpnT), nullptr, nullptr) + 256;
}

// numpy MT19937 seeding (init_genrand via _legacy_seeding(seed)): key[624] + pos = 624
extern "C" int sd_mt19937_seed_host(uint32_t seed, int32_t* state_host) {
  SD_REQUIRE(state_host, "state_host is null");
  uint32_t k = seed;
  for (int i = 0; i < 624; ++i) {
    state_host[i] = (int32_t)k;
    k = 1812433253u * (k ^ (k >> 30)) + (uint32_t)(i + 1);
  }
  state_host[624] = 624;
  return SD_OK;
}

extern "C" int sd_rpn_anchor_target(const float* im_info, const float* gt_bbox, int B, int M, int G,
                                    const sd_rpn_target_param* param_host, int32_t* mt_state,
                                    float* cls_label, float* reg_target, float* reg_weight,
                                    int layout, void* workspace, size_t workspace_bytes,
                                    void* stream) {
  SD_REQUIRE(param_host, "param is null");
  RpnArgs a{};
  a.p = *param_host;
  if (int e = rpn_count(a.p, &a.A, &a.N, &a.sumHW)) return e;
  SD_REQUIRE(B >= 0 && M >= 0 && (G == 4 || G == 5), "bad B / M / gt row width (4 or 5)");
  SD_REQUIRE(layout == 0 || layout == 1, "layout must be 0 (flat) or 1 (loader layout)");
  SD_REQUIRE(a.p.image_anchor >= 0 && a.p.image_anchor <= kRpnMaxKeep, "image_anchor outside [0,%d]",
             kRpnMaxKeep);
  SD_REQUIRE(a.p.pos_fraction >= 0.0 && a.p.pos_fraction <= 1.0, "pos_fraction outside [0,1]");
  if (B == 0) return SD_OK;
  SD_REQUIRE(im_info && (gt_bbox || M == 0) && mt_state && cls_label && reg_target && reg_weight,
             "null pointer");
  SD_REQUIRE((((uintptr_t)reg_target | (uintptr_t)reg_weight) & 15) == 0, "outputs must be 16-B aligned");
  // num_fg = int(fg_fraction * num) with a python float fraction: evaluated in double
  a.num_fg = (int)(a.p.pos_fraction * (double)a.p.image_anchor);
  // base anchors: core/detection_input.py:373-399 in double (np.round = rint, half to even)
  for (int l = 0; l < a.p.nlvl; ++l) {
    const double s = a.p.stride[l];
    const double w = s, h = s, x_ctr = 0.5 * (w - 1), y_ctr = 0.5 * (h - 1);
    for (int i = 0; i < a.p.n_aspects; ++i) {
      const double wr = rint(sqrt(w * h / a.p.aspects[i]));
      const double hr = rint(wr * a.p.aspects[i]);
      for (int j = 0; j < a.p.n_scales; ++j) {
        const double ws = wr * a.p.scales[j], hs = hr * a.p.scales[j];
        float* b = a.base[l][i * a.p.n_scales + j];
        b[0] = (float)(x_ctr - 0.5 * (ws - 1));
        b[1] = (float)(y_ctr - 0.5 * (hs - 1));
        b[2] = (float)(x_ctr + 0.5 * (ws - 1));
        b[3] = (float)(y_ctr + 0.5 * (hs - 1));
      }
    }
  }
  a.im_info = im_info; a.gt = gt_bbox; a.B = B; a.M = M; a.G = G; a.mt = mt_state;
  a.cls = cls_label; a.tgt = reg_target; a.wgt = reg_weight; a.layout = layout;
  a.nblk = cdiv(a.N, kRpnT);
  char* base = reinterpret_cast<char*>(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
  const size_t need = rpn_layout(B, a.N, M, a.nblk, &a, base) + (size_t)(base - (char*)workspace);
  if (!workspace || workspace_bytes < need)
    return fail(SD_ERR_WORKSPACE, "rpn_anchor_target workspace too small: %zu < %zu bytes",
                workspace_bytes, need);
  hipStream_t st = (hipStream_t)stream;
  SD_HIP_CHECK(hipMemsetAsync(a.gtmax, 0, (size_t)B * (M > 0 ? M : 1) * 4, st));
  const size_t lds = (size_t)(M > 0 ? M : 1) * (sizeof(float4) + sizeof(unsigned));
  SD_REQUIRE(lds <= 64 * 1024, "too many gt boxes per image (M=%d)", M);
  const dim3 grid(a.nblk, B);
  hipLaunchKernelGGL(rpn_overlap_kernel, grid, dim3(kRpnT), lds, st, a);
  hipLaunchKernelGGL(rpn_label_kernel, grid, dim3(kRpnT), lds, st, a);
  hipLaunchKernelGGL(rpn_scan_kernel, dim3(B), dim3(1024), 0, st, a);
  hipLaunchKernelGGL(rpn_lists_kernel, grid, dim3(kRpnT), 0, st, a);
  hipLaunchKernelGGL(rpn_sample_kernel, dim3(1), dim3((1 + kMtProd) * kWave), 0, st, a);
  hipLaunchKernelGGL(rpn_encode_kernel, grid, dim3(kRpnT), 0, st, a);
  SD_LAUNCH_CHECK();
  return SD_OK;
}
