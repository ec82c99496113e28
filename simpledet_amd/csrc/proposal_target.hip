// ProposalTarget for gfx950, device resident (the reference copies rois/gt to the host, runs a
// single-threaded loop and copies five tensors back, stalling the stream every training step).
//   reference: operator_cxx/proposal_target-inl.h:123-256 (Forward), proposal_target.cc:21-163
//              (SampleROI), :165-185 (BBoxOverlap), :187-202 (ExpandBboxRegressionTargets),
//              :204-227 (NonLinearTransformAndNormalization); RNG = std::random_shuffle over libc
//              rand() (glibc TYPE_3 additive feedback generator).
// Three launches on one stream, no host involvement:
//   1. pt_assign_kernel   one workgroup per image: order-preserving compaction of the valid gt
//                         boxes and of the rois with y2 > 0 (+ appended gt), IoU of every candidate
//                         against the gt boxes held in LDS, first-maximum arg-max, then the fg / neg
//                         / bg index lists in candidate order (block-wide prefix scans).
//   2. pt_sample_kernel   ONE workgroup, images in order: replays the reference's random_shuffle
//                         calls bit for bit from the glibc state kept in device memory (the draw
//                         count is data dependent, so the state has to live where the data is):
//                         draws from a register-resident ring, j_i = draw % (i+1) in parallel,
//                         and the kept prefix by walking the swaps backwards, one thread per row.
//   3. pt_encode_kernel   one workgroup per output row: gather roi / label / IoU, encode the box
//                         deltas, write the 4-of-4K expanded target and weight rows (zero fill +
//                         slot) with coalesced stores.
// Float parity: IoU and the delta arithmetic use the reference's operation order with IEEE divide
// (bit exact); log() is the device logf, within 1 ulp of glibc logf (tests: 1e-6 relative).
#include "common.h"
#include "../../include/simpledet_ops.h"
#include <math.h>

namespace sd {

struct PtWs {
  float4* cand;     // (B, Ncand) candidate boxes
  float* ov;        // (B, Ncand) max overlap
  int* gta;         // (B, Ncand) arg-max gt (index into the valid-gt list)
  float4* gtbox;    // (B, M) valid gt boxes, compacted
  float* gtcls;     // (B, M)
  int* fg;          // (B, Ncand)
  int* neg;         // (B, Ncand)
  int* bg;          // (B, Ncand)
  int* counts;      // (B, 8): n_cand, n_gt, n_fg, n_neg, n_bg, fg_this, n_kept, no_gt
  int* kept;        // (B, S)
  int* gtsrc;       // (B, M) input row of every valid gt box (ProposalMaskTarget: its polygon)
};

struct PtArgs {
  const float* rois;
  const float* gt;
  const float* valid_ranges;  // ProposalTarget_v2 / MaskTarget: (B,2) [min, max] object scale, or null
  int filter_scales;          // append only the gt boxes inside the valid range
  int v2;                     // v2 / MaskTarget: an empty roi / gt list is replaced by one zero row
  const float* polys;         // ProposalMaskTarget: (B, M, L) gt polygons, or null
  float* mask_out;            // (B, FG, ms, ms)
  float* ratio_out;           // output_ratio: (B, FG) mask ratio, or null
  unsigned long long* bits;   // output_ratio: per fg row 2 x bit_words words (toggle / union bitmaps)
  int bit_words;              // words per bitmap = ceil(max_raster_pixels / 64)
  int L, mask_size, FG;
  PtWs ws;
  sd_proposal_target_param p;
  int B, N, M, Mws, Ncand, S, fg_per_image;  // Mws = max(M, 1): row stride of the gt workspace
  int* rng;
  float* roi_out;
  float* label;
  float* bbox_target;
  float* bbox_weight;
  float* iou_out;
  int* kept_index;
};

// block-wide exclusive prefix sum of a 0/1 flag in thread order; returns this thread's offset and
// the block total (all threads must call)
template <int THREADS>
__device__ __forceinline__ int block_scan_flag(bool flag, int* wave_sums, int* total) {
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const unsigned long long b = __ballot(flag);
  const int within = __popcll(b & ((1ull << lane) - 1));
  __syncthreads();  // wave_sums may still be read from a previous call
  if (lane == 0) wave_sums[wave] = __popcll(b);
  __syncthreads();
  int off = 0, tot = 0;
  for (int w = 0; w < THREADS / kWave; ++w) {
    const int v = wave_sums[w];
    if (w < wave) off += v;
    tot += v;
  }
  *total = tot;
  return off + within;
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void pt_assign_kernel(PtArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ int wave_sums[THREADS / kWave];
  const int img = blockIdx.x, tid = threadIdx.x;
  float4* gbox = reinterpret_cast<float4*>(smem);  // M boxes
  float* garea = smem + 4 * a.M;                   // M areas
  PtWs ws = a.ws;
  float4* cand = ws.cand + (long)img * a.Ncand;
  float4* gtbox = ws.gtbox + (long)img * a.Mws;
  float* gtcls = ws.gtcls + (long)img * a.Mws;
  int* gtsrc = ws.gtsrc + (long)img * a.Mws;

  // ---- valid gt boxes (cls != -1), order preserved (proposal_target-inl.h:155-162) ----
  int n_gt = 0;
  for (int base = 0; base < a.M; base += THREADS) {
    const int j = base + tid;
    float g[5] = {0, 0, 0, 0, -1};
    if (j < a.M)
      for (int k = 0; k < 5; ++k) g[k] = a.gt[((long)img * a.M + j) * 5 + k];
    const bool ok = j < a.M && g[4] != -1.f;
    int tot;
    const int off = block_scan_flag<THREADS>(ok, wave_sums, &tot);
    if (ok) {
      const float4 b = make_float4(g[0], g[1], g[2], g[3]);
      gbox[n_gt + off] = b;
      garea[n_gt + off] = (b.z - b.x + 1.f) * (b.w - b.y + 1.f);
      gtbox[n_gt + off] = b;
      gtcls[n_gt + off] = g[4];
      gtsrc[n_gt + off] = j;
    }
    n_gt += tot;
  }
  // ---- candidate rois: y2 > 0 in order (:171-175), then every valid gt box (:177-185) ----
  int n_cand = 0;
  for (int base = 0; base < a.N; base += THREADS) {
    const int j = base + tid;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (j < a.N) r = reinterpret_cast<const float4*>(a.rois)[(long)img * a.N + j];
    const bool ok = j < a.N && r.w > 0.f;
    int tot;
    const int off = block_scan_flag<THREADS>(ok, wave_sums, &tot);
    if (ok) cand[n_cand + off] = r;
    n_cand += tot;
  }
  __syncthreads();
  if (!a.p.proposal_without_gt) {
    if (a.valid_ranges && a.filter_scales) {
      // ProposalTarget_v2 (proposal_target_v2-inl.h:188-203): a gt box joins the candidates only if
      // its area lies in [valid_min^2, valid_max^2]; order preserved
      const float v0 = a.valid_ranges[2 * img], v1 = a.valid_ranges[2 * img + 1];
      const float vmin = v0 * v0, vmax = v1 * v1;
      for (int base = 0; base < n_gt; base += THREADS) {
        const int j = base + tid;
        bool ok = false;
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (j < n_gt) {
          b = gbox[j];
          const float w = (float)((double)(b.z - b.x) + 1.0), h = (float)((double)(b.w - b.y) + 1.0);
          ok = !(w * h < vmin || w * h > vmax);
        }
        int tot;
        const int off = block_scan_flag<THREADS>(ok, wave_sums, &tot);
        if (ok) cand[n_cand + off] = b;
        n_cand += tot;
      }
    } else {
      for (int j = tid; j < n_gt; j += THREADS) cand[n_cand + j] = gbox[j];
      n_cand += n_gt;
    }
  }
  if (a.v2) {
    // v2 (:244-249): no candidate -> one all-zero roi; no valid gt -> one all-zero gt row (its
    // class column is read out of bounds by the reference; defined as 0 here)
    if (n_cand == 0) {
      if (tid == 0) cand[0] = make_float4(0.f, 0.f, 0.f, 0.f);
      n_cand = 1;
    }
    if (n_gt == 0) {
      if (tid == 0) {
        gbox[0] = make_float4(0.f, 0.f, 0.f, 0.f);
        garea[0] = 1.f;
        gtbox[0] = gbox[0];
        gtcls[0] = 0.f;
        gtsrc[0] = -1;  // no polygon: ProposalMaskTarget draws an empty mask
      }
      n_gt = 1;
    }
  }
  __syncthreads();  // cand[] written by other threads is read below (same workgroup: L1/L2 coherent)
  __threadfence_block();

  // ---- IoU (proposal_target.cc:165-185) + first-maximum arg-max (:47-61) + lists (:67-104) ----
  float* ov = ws.ov + (long)img * a.Ncand;
  int* gta = ws.gta + (long)img * a.Ncand;
  int* fg = ws.fg + (long)img * a.Ncand;
  int* neg = ws.neg + (long)img * a.Ncand;
  int* bg = ws.bg + (long)img * a.Ncand;
  int n_fg = 0, n_neg = 0, n_bg = 0;
  for (int base = 0; base < n_cand; base += THREADS) {
    const int i = base + tid;
    float maxv = 0.f;
    int maxi = 0;
    if (i < n_cand && n_gt > 0) {
      const float4 b = cand[i];
      const float box_area = (b.z - b.x + 1.f) * (b.w - b.y + 1.f);
      for (int j = 0; j < n_gt; ++j) {
        const float4 q = gbox[j];
        float o = 0.f;
        // std::min(a, b) = b < a ? b : a ; std::max(a, b) = a < b ? b : a
        const float iw = (q.z < b.z ? q.z : b.z) - (b.x < q.x ? q.x : b.x) + 1.f;
        if (iw > 0) {
          const float ih = (q.w < b.w ? q.w : b.w) - (b.y < q.y ? q.y : b.y) + 1.f;
          if (ih > 0) {
            const float union_area = box_area + garea[j] - iw * ih;
            o = iw * ih / union_area;
          }
        }
        if (j == 0) maxv = o;
        else if (maxv < o) {
          maxv = o;
          maxi = j;
        }
      }
    }
    if (i < n_cand) {
      ov[i] = maxv;
      gta[i] = maxi;
    }
    const bool valid = i < n_cand;
    const bool is_fg = valid && maxv >= a.p.fg_thresh;
    const bool is_neg = valid && !is_fg;
    const bool is_bg = valid && maxv >= a.p.bg_thresh_lo && maxv < a.p.bg_thresh_hi;
    int tot, off;
    off = block_scan_flag<THREADS>(is_fg, wave_sums, &tot);
    if (is_fg) fg[n_fg + off] = i;
    n_fg += tot;
    off = block_scan_flag<THREADS>(is_neg, wave_sums, &tot);
    if (is_neg) neg[n_neg + off] = i;
    n_neg += tot;
    off = block_scan_flag<THREADS>(is_bg, wave_sums, &tot);
    if (is_bg) bg[n_bg + off] = i;
    n_bg += tot;
  }
  if (tid == 0) {
    int* c = ws.counts + img * 8;
    c[0] = n_cand; c[1] = n_gt; c[2] = n_fg; c[3] = n_neg; c[4] = n_bg;
  }
}

// ---- std::random_shuffle replay, parallel where the algorithm allows it -------------------------
// random_shuffle(first, last):  for i in [1, n): swap(a[i], a[rand() % (i + 1)]).
// (1) the n-1 draws do not depend on the data: one lane generates them with the 31-word ring held
//     in registers.  In coordinates rotated so that the front index is 0 the rear index is always
//     28 (TYPE_3 keeps front = rear + 3), so the unrolled recurrence reg[s] += reg[(s+28) % 31]
//     has static register indices: ~3 instructions per draw instead of a chain of LDS round trips;
// (2) j_i = draw_i % (i+1) for all i in parallel;
// (3) only the first `take` shuffled entries are ever used.  Output position p is decided by the
//     LAST swap from above that writes it (last[p] = max i with j_i == p, one LDS atomic-max pass
//     over the swaps): it holds list[last[p]], untouched by any earlier swap.  Positions no later
//     swap writes (probability (p+1)/n) keep what step p left: those walk the swaps i <= p
//     backwards (at i == cur the content came from j_i; at j_i == cur it came from i), i.e. at most
//     `take` steps instead of n.  One thread per output.
// The rare multi-round negative padding (proposal_target.cc:116-122 with fewer negatives than
// missing rows) needs the whole permuted list between rounds and replays the swaps in order.
struct RingRegs {
  unsigned r[31];
};

template <int THREADS>
__device__ void draw_many(unsigned* ring, int* fb, int* draws, int n) {
  // every lane of wave 0 runs the same recurrence on the same values; the stores of a draw all
  // carry the same value to the same LDS word (no exec-mask juggling per draw)
  if (threadIdx.x < kWave) {
    const int f = fb[0];
    RingRegs g;
#pragma unroll
    for (int k = 0; k < 31; ++k) g.r[k] = ring[(f + k) % 31];
    int base = 0;
    for (; base + 31 <= n; base += 31) {
#pragma unroll
      for (int s = 0; s < 31; ++s) {
        g.r[s] += g.r[(s + 28) % 31];
        draws[base + s] = (int)(g.r[s] >> 1);
      }
    }
#pragma unroll
    for (int s = 0; s < 31; ++s) {
      if (base + s < n) {
        g.r[s] += g.r[(s + 28) % 31];
        draws[base + s] = (int)(g.r[s] >> 1);
      }
    }
    if (threadIdx.x == 0) {
#pragma unroll
      for (int k = 0; k < 31; ++k) ring[(f + k) % 31] = g.r[k];
      fb[0] = (f + n) % 31;
      fb[1] = (fb[1] + n) % 31;
    }
  }
  __syncthreads();
}

// in-order replay (multi-round padding only): lane 0, list in LDS
__device__ void shuffle_list(int* list, int n, unsigned* ring, int& f, int& b) {
  for (int i = 1; i < n; ++i) {
    const unsigned v = (ring[f] += ring[b]);
    if (++f >= 31) f = 0;
    if (++b >= 31) b = 0;
    const int j = (int)(v >> 1) % (i + 1);
    if (i != j) {
      const int t = list[i];
      list[i] = list[j];
      list[j] = t;
    }
  }
}

template <int THREADS>
__global__ __launch_bounds__(THREADS) void pt_sample_kernel(PtArgs a) {
  extern __shared__ __attribute__((aligned(16))) int lds[];  // [list: Ncand][draws / j: Ncand][last: Ncand]
  __shared__ unsigned ring[31];
  __shared__ int fb[2];
  const int tid = threadIdx.x;
  int* lists = lds;
  int* draws = lds + a.Ncand;
  int* last = lds + 2 * a.Ncand;  // last[k] = the largest i with j_i == k (0: none)
  if (tid < 31) ring[tid] = (unsigned)a.rng[tid];
  if (tid == 0) {
    fb[0] = a.rng[31];
    fb[1] = a.rng[32];
  }
  __syncthreads();
  const int S = a.S;
  // out[0..take) = first `take` entries of random_shuffle(list[0..n)); consumes n-1 draws
  auto shuffled_prefix = [&](const int* list, int n, int take, int* out) {
    draw_many<THREADS>(ring, fb, draws, n - 1);
    for (int i = tid; i < n; i += THREADS) last[i] = 0;
    __syncthreads();
    for (int i = 1 + tid; i < n; i += THREADS) {
      const int j = draws[i - 1] % (i + 1);  // j_i
      draws[i - 1] = j;
      if (j != i) atomicMax(&last[j], i);    // the last swap that writes position j from above
    }
    __syncthreads();
    for (int p = tid; p < take; p += THREADS) {
      // If some later swap (i > p, j_i == p) writes position p, the last of them decides: it
      // brings list[i], which no earlier swap can have touched.  Otherwise position p holds what
      // step p left there, and only the swaps i <= p matter: walk those backwards (at i == cur
      // the content came from j_i; at j_i == cur it came from i, which ends the walk).
      const int L = last[p];
      int cur = p;
      if (L > p) {
        cur = L;
      } else {
        // j_i are read eight at a time (independent LDS loads); once cur has moved up to some i
        // no later (smaller) i can match it, so the walk simply runs on without a branch
        int i = p;
        for (; i >= 8; i -= 8) {
          int j[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) j[e] = draws[i - 1 - e];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int ii = i - e;
            cur = (cur == ii) ? j[e] : ((cur == j[e]) ? ii : cur);
          }
        }
        for (; i >= 1; --i) {
          const int j = draws[i - 1];
          cur = (cur == i) ? j : ((cur == j) ? i : cur);
        }
      }
      out[p] = list[cur];
    }
    __syncthreads();
  };
  for (int img = 0; img < a.B; ++img) {
    PtWs ws = a.ws;
    int* c = ws.counts + img * 8;
    const int n_fg = c[2], n_neg = c[3], n_bg = c[4];
    int* fg = ws.fg + (long)img * a.Ncand;
    int* neg = ws.neg + (long)img * a.Ncand;
    int* bg = ws.bg + (long)img * a.Ncand;
    int* kept = ws.kept + (long)img * S;
    const int fg_this = a.fg_per_image < n_fg ? a.fg_per_image : n_fg;       // :81
    const int want_bg = S - fg_this;
    const int bg_this = want_bg < n_bg ? want_bg : n_bg;                      // :100
    int nkept = 0;
    if (n_fg > fg_this) {                                                     // :82-85
      shuffled_prefix(fg, n_fg, fg_this, kept);
    } else {
      for (int i = tid; i < fg_this; i += THREADS) kept[i] = fg[i];
    }
    nkept = fg_this;
    if (n_bg > bg_this) {                                                     // :101-104
      shuffled_prefix(bg, n_bg, bg_this, kept + nkept);
    } else {
      for (int i = tid; i < bg_this; i += THREADS) kept[nkept + i] = bg[i];
    }
    nkept += bg_this;
    // pad with shuffled negatives (:116-122)
    if (nkept < S && n_neg > 0) {
      const int gap = S - nkept;
      if (n_neg >= gap) {  // one round: only its first `gap` entries matter
        shuffled_prefix(neg, n_neg, gap, kept + nkept);
        nkept += gap;
      } else {             // several rounds over the same, progressively re-shuffled list
        for (int i = tid; i < n_neg; i += THREADS) lists[i] = neg[i];
        __syncthreads();
        while (nkept < S) {
          const int g2 = S - nkept;
          if (tid == 0) {
            int f = fb[0], b = fb[1];
            shuffle_list(lists, n_neg, ring, f, b);
            fb[0] = f;
            fb[1] = b;
          }
          __syncthreads();
          const int take = g2 < n_neg ? g2 : n_neg;
          for (int i = tid; i < take; i += THREADS) kept[nkept + i] = lists[i];
          nkept += take;
          __syncthreads();
        }
      }
    }
    for (int i = nkept + tid; i < S; i += THREADS) kept[i] = -1;
    if (tid == 0) {
      c[5] = fg_this;
      c[6] = nkept;
    }
    __syncthreads();
  }
  if (tid < 31) a.rng[tid] = (int)ring[tid];
  if (tid == 0) {
    a.rng[31] = fb[0];
    a.rng[32] = fb[1];
  }
}

__global__ __launch_bounds__(128) void pt_encode_kernel(PtArgs a) {
  const int row = blockIdx.x;  // (img, slot)
  const int img = row / a.S, slot = row % a.S;
  const int tid = threadIdx.x;
  PtWs ws = a.ws;
  const int* c = ws.counts + img * 8;
  const int n_gt = c[1], fg_this = c[5], nkept = c[6];
  const int K4 = 4 * a.p.num_classes;
  float* bt = a.bbox_target + (long)row * K4;
  float* bw = a.bbox_weight + (long)row * K4;
  float4 roi = make_float4(0.f, 0.f, 0.f, 0.f);
  float lab = 0.f, iou = 0.f;
  int start = -1;
  float t[4] = {0.f, 0.f, 0.f, 0.f};
  int kidx = -1;
  if (slot < nkept) {
    kidx = ws.kept[(long)img * a.S + slot];
    roi = ws.cand[(long)img * a.Ncand + kidx];
    iou = ws.ov[(long)img * a.Ncand + kidx];
    if (n_gt > 0) {
      const int g = ws.gta[(long)img * a.Ncand + kidx];
      if (slot < fg_this) lab = ws.gtcls[(long)img * a.Mws + g];   // proposal_target.cc:129-131
      const float4 gt = ws.gtbox[(long)img * a.Mws + g];
      // NonLinearTransformAndNormalization :204-227 ("0.5 *" is a double expression)
      const float ex_width = roi.z - roi.x + 1.f;
      const float ex_height = roi.w - roi.y + 1.f;
      const float ex_ctr_x = (float)((double)roi.x + 0.5 * (double)(ex_width - 1.f));
      const float ex_ctr_y = (float)((double)roi.y + 0.5 * (double)(ex_height - 1.f));
      const float gt_width = gt.z - gt.x + 1.f;
      const float gt_height = gt.w - gt.y + 1.f;
      const float gt_ctr_x = (float)((double)gt.x + 0.5 * (double)(gt_width - 1.f));
      const float gt_ctr_y = (float)((double)gt.y + 0.5 * (double)(gt_height - 1.f));
      t[0] = (gt_ctr_x - ex_ctr_x) / (ex_width + 1e-14f);
      t[1] = (gt_ctr_y - ex_ctr_y) / (ex_height + 1e-14f);
      t[2] = logf(gt_width / ex_width);
      t[3] = logf(gt_height / ex_height);
#pragma unroll
      for (int k = 0; k < 4; ++k) t[k] = (t[k] - a.p.bbox_mean[k]) / a.p.bbox_std[k];
      float cls_f = lab;
      if (a.p.class_agnostic) cls_f = lab < 1.f ? lab : 1.f;     // :151-154
      if (cls_f > 0.f) {                                          // :193-200
        const int cls = (int)cls_f;
        if (4 * cls + 4 <= K4) start = 4 * cls;
      }
    }
  }
  for (int k = tid; k < K4; k += blockDim.x) {
    float tv = 0.f, wv = 0.f;
    if (start >= 0 && k >= start && k < start + 4) {
      tv = t[k - start];
      wv = a.p.bbox_weight[k - start];
    }
    bt[k] = tv;
    bw[k] = wv;
  }
  if (tid == 0) {
    reinterpret_cast<float4*>(a.roi_out)[row] = roi;
    a.label[row] = lab;
    a.iou_out[row] = iou;
    if (a.kept_index) a.kept_index[row] = kidx;
  }
}


// ---- ProposalMaskTarget: polygon of the assigned gt box -> mask_size x mask_size target -----------
// reference: convertPoly2Mask, operator_cxx/proposal_mask_target.cc:148-216, over the COCO mask API
// (rleFrPoly + rleDecode of github.com/RogerChern/cocoapi, un-vendored: the published pycocotools
// algorithm is followed -- parity unpinned, see oracle/mask_api.c).
// One workgroup per (image, foreground slot).  rleFrPoly is a sequential walk; its result, though,
// is "pixel t (column-major) is set iff an odd number of boundary crossings has position <= t", so
//   * every vertex is scaled x5 and rounded, every edge owns max(|dx|,|dy|)+1 dense points (block
//     prefix sum over the edges),
//   * every dense point is handled by its own thread: it recomputes itself and its predecessor
//     with the reference's double expressions, and where the up-sampled column changes it toggles
//     one bit of an LDS bitmap at x*h + ceil(y) (two crossings at one place cancel, exactly like
//     the zero-length runs rleFrPoly merges),
//   * one wave turns the bitmap into the mask with a ballot/popcount prefix parity,
//   * the segments of a polygon are OR-ed (:196-206).
// The reference swaps x and y on the way in, which makes the column-major decode row-major.
constexpr int kMaskThreads = 256;
constexpr int kMaskMaxVerts = 2048;  // vertices per segment held in LDS

__device__ __forceinline__ void poly_point(const int* vx, const int* vy, int e, int d, int* u, int* v) {
  int xs = vx[e], xe = vx[e + 1], ys = vy[e], ye = vy[e + 1];
  const int dx = abs(xe - xs), dy = abs(ys - ye);
  const bool flip = (dx >= dy && xs > xe) || (dx < dy && ys > ye);
  if (flip) { int t = xs; xs = xe; xe = t; t = ys; ys = ye; ye = t; }
  const double s = dx >= dy ? (double)(ye - ys) / (double)dx : (double)(xe - xs) / (double)dy;
  if (dx >= dy) {
    const int t = flip ? dx - d : d;
    *u = t + xs;
    *v = (int)((double)ys + s * (double)t + .5);
  } else {
    const int t = flip ? dy - d : d;
    *v = t + ys;
    *u = (int)((double)xs + s * (double)t + .5);
  }
}

__global__ __launch_bounds__(kMaskThreads) void pt_mask_kernel(PtArgs a) {
  extern __shared__ __attribute__((aligned(16))) int msm[];
  __shared__ int wave_sums[kMaskThreads / kWave];
  __shared__ int s_total;
  const int row = blockIdx.x, img = row / a.FG, slot = row % a.FG, tid = threadIdx.x;
  const int ms = a.mask_size, area = ms * ms;
  float* out = a.mask_out + (long)row * area;
  const int* c = a.ws.counts + img * 8;
  const int fg_this = c[5];
  if (slot >= fg_this) {  // rows past the sampled foreground keep the -1 fill (-inl.h:211-212)
    for (int j = tid; j < area; j += kMaskThreads) out[j] = -1.f;
    return;
  }
  int* vx = msm;                      // [kMaskMaxVerts + 1]
  int* vy = vx + kMaskMaxVerts + 1;   // [kMaskMaxVerts + 1]
  int* start = vy + kMaskMaxVerts + 1;  // [kMaskMaxVerts + 1] first dense point of every edge
  int* tog = start + kMaskMaxVerts + 1;  // [area] crossing parity per position
  int* acc = tog + area;                 // [area] OR over the segments
  for (int j = tid; j < area; j += kMaskThreads) { tog[j] = 0; acc[j] = 0; }
  const int kidx = a.ws.kept[(long)img * a.S + slot];
  const int g = a.ws.gta[(long)img * a.Ncand + kidx];
  const int src = a.ws.gtsrc[(long)img * a.Mws + g];
  const float4 roi = reinterpret_cast<const float4*>(a.roi_out)[(long)img * a.S + slot];
  float w = roi.z - roi.x, h = roi.w - roi.y;
  w = 1.f < w ? w : 1.f;  // std::max((DType)1., w)
  h = 1.f < h ? h : 1.f;
  const float* poly = src >= 0 ? a.polys + ((long)img * a.M + src) * a.L : nullptr;
  int n_seg = poly ? (int)poly[1] : 0;
  if (n_seg < 0) n_seg = 0;
  int offset = 2 + n_seg;
  __syncthreads();
  for (int sgi = 0; sgi < n_seg; ++sgi) {
    const int cur_len = (int)poly[sgi + 2];
    int k = cur_len / 2;
    if (k > kMaskMaxVerts) k = kMaskMaxVerts;  // (the reference has no limit; L bounds it in practice)
    // vertices in the RoI's mask frame, x5, rounded: "x" of the mask API = row coordinate
    for (int t = tid; t < k; t += kMaskThreads) {
      const float px = poly[offset + 2 * t], py = poly[offset + 2 * t + 1];
      if (a.ratio_out) {  // convertPoly2MaskWithRatio: `double poly_index` (:53-63)
        const double fy = ((double)py - (double)roi.y) * (double)ms / (double)h;
        const double fx = ((double)px - (double)roi.x) * (double)ms / (double)w;
        vx[t] = (int)(5.0 * fy + .5);
        vy[t] = (int)(5.0 * fx + .5);
      } else {
        const float fy = (py - roi.y) * (float)ms / h;  // xys[2t]
        const float fx = (px - roi.x) * (float)ms / w;  // xys[2t+1]
        vx[t] = (int)(5.0 * (double)fy + .5);
        vy[t] = (int)(5.0 * (double)fx + .5);
      }
    }
    __syncthreads();
    if (tid == 0 && k > 0) { vx[k] = vx[0]; vy[k] = vy[0]; }
    __syncthreads();
    // dense points per edge, exclusive prefix sum
    int run = 0;
    for (int base = 0; base < k; base += kMaskThreads) {
      const int e = base + tid;
      int cnt = 0;
      if (e < k) {
        const int dx = abs(vx[e] - vx[e + 1]), dy = abs(vy[e] - vy[e + 1]);
        cnt = (dx > dy ? dx : dy) + 1;
      }
      // inclusive scan inside the wave, then across the waves
      int incl = cnt;
#pragma unroll
      for (int o = 1; o < kWave; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if ((tid & (kWave - 1)) >= o) incl += t;
      }
      __syncthreads();
      if ((tid & (kWave - 1)) == kWave - 1) wave_sums[tid / kWave] = incl;
      __syncthreads();
      int woff = 0, tot = 0;
      for (int wv = 0; wv < kMaskThreads / kWave; ++wv) {
        if (wv < tid / kWave) woff += wave_sums[wv];
        tot += wave_sums[wv];
      }
      if (e < k) start[e] = run + woff + incl - cnt;
      run += tot;
    }
    if (tid == 0) { start[k] = run; s_total = run; }
    __syncthreads();
    const int m = s_total;
    for (int gidx = 1 + tid; gidx < m; gidx += kMaskThreads) {
      // edge of this dense point: last e with start[e] <= gidx
      int lo = 0, hi = k - 1;
      while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (start[mid] <= gidx) lo = mid; else hi = mid - 1;
      }
      int u1, v1, u0, v0;
      poly_point(vx, vy, lo, gidx - start[lo], &u1, &v1);
      if (gidx - 1 >= start[lo]) poly_point(vx, vy, lo, gidx - 1 - start[lo], &u0, &v0);
      else poly_point(vx, vy, lo - 1, gidx - 1 - start[lo - 1], &u0, &v0);
      if (u1 != u0) {
        double xd = (double)(u1 < u0 ? u1 : u1 - 1);
        xd = (xd + .5) / 5.0 - .5;
        if (floor(xd) != xd || xd < 0 || xd > (double)(ms - 1)) continue;
        double yd = (double)(v1 < v0 ? v1 : v0);
        yd = (yd + .5) / 5.0 - .5;
        if (yd < 0) yd = 0; else if (yd > (double)ms) yd = (double)ms;
        yd = ceil(yd);
        const int pos = (int)xd * ms + (int)yd;
        if (pos < area) atomicXor(&tog[pos], 1);
      }
    }
    __syncthreads();
    if (tid < kWave) {  // prefix parity -> mask of this segment, OR-ed into acc
      int carry = 0;
      for (int base = 0; base < area; base += kWave) {
        const int pos = base + tid;
        const int bit = pos < area ? tog[pos] : 0;
        const unsigned long long bits = __ballot(bit);
        const int par = (__popcll(bits & ((2ull << tid) - 1ull)) + carry) & 1;
        if (pos < area) {
          if (par) acc[pos] = 1;
          tog[pos] = 0;
        }
        carry ^= __popcll(bits) & 1;
      }
    }
    __syncthreads();
    offset += cur_len;
  }
  for (int j = tid; j < area; j += kMaskThreads) out[j] = acc[j] ? 1.f : 0.f;
}

// ------------------------------------------------------------------------------------------------
// output_ratio (mask scoring R-CNN, models/msrcnn/builder.py:219-237): convertPoly2MaskWithRatio,
// proposal_mask_target.cc:20-152.  Besides the mask, the polygon is rasterised at image resolution
// inside the RoI (crop_h x crop_w, corners truncated to int) and inside the bounding box of RoI and
// polygon (full_h x full_w); the ratio is crop pixels / (full pixels + 1e-4), at least 1e-10.
// Those rasters are up to the image size, so the crossing bitmap of pt_mask_kernel moves to global
// memory, one bit per pixel: toggles by 64-bit atomic xor, and every lane owns a contiguous run of
// words for zeroing, the prefix parity (word parities scanned across the workgroup, a shift-xor
// ladder inside the word), the OR over the segments and the final popcount.  A raster larger than
// the caller's bound (max_raster_pixels) yields NaN for that row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kMaskThreads) void pt_mask_ratio_kernel(PtArgs a) {
  __shared__ int vx[kMaskMaxVerts + 1], vy[kMaskMaxVerts + 1], start[kMaskMaxVerts + 1];
  __shared__ int wave_sums[kMaskThreads / kWave];
  __shared__ int s_total;
  __shared__ double s_red[4][kMaskThreads / kWave];
  __shared__ int s_par[kMaskThreads];
  const int row = blockIdx.x, img = row / a.FG, slot = row % a.FG, tid = threadIdx.x;
  const int* c = a.ws.counts + img * 8;
  const int fg_this = c[5];
  if (slot >= fg_this) {  // -inl.h:244: zero filled
    if (tid == 0) a.ratio_out[row] = 0.f;
    return;
  }
  const int kidx = a.ws.kept[(long)img * a.S + slot];
  const int g = a.ws.gta[(long)img * a.Ncand + kidx];
  const int src = a.ws.gtsrc[(long)img * a.Mws + g];
  const float4 roi = reinterpret_cast<const float4*>(a.roi_out)[(long)img * a.S + slot];
  const float* poly = src >= 0 ? a.polys + ((long)img * a.M + src) * a.L : nullptr;
  int n_seg = poly ? (int)poly[1] : 0;
  if (n_seg < 0) n_seg = 0;
  // :46-66 bounding box of RoI and polygon, in double
  double bx1 = (double)roi.x, bx2 = (double)roi.z, by1 = (double)roi.y, by2 = (double)roi.w;
  {
    int offset = 2 + n_seg;
    for (int sgi = 0; sgi < n_seg; ++sgi) {
      const int cur_len = (int)poly[sgi + 2];
      for (int t = tid; t < cur_len / 2; t += kMaskThreads) {
        const double px = (double)poly[offset + 2 * t], py = (double)poly[offset + 2 * t + 1];
        bx1 = px < bx1 ? px : bx1; bx2 = px > bx2 ? px : bx2;
        by1 = py < by1 ? py : by1; by2 = py > by2 ? py : by2;
      }
      offset += cur_len;
    }
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) {
      const double t1 = __shfl_xor(bx1, o), t2 = __shfl_xor(bx2, o), t3 = __shfl_xor(by1, o), t4 = __shfl_xor(by2, o);
      bx1 = t1 < bx1 ? t1 : bx1; bx2 = t2 > bx2 ? t2 : bx2;
      by1 = t3 < by1 ? t3 : by1; by2 = t4 > by2 ? t4 : by2;
    }
    if ((tid & (kWave - 1)) == 0) {
      s_red[0][tid / kWave] = bx1; s_red[1][tid / kWave] = bx2;
      s_red[2][tid / kWave] = by1; s_red[3][tid / kWave] = by2;
    }
    __syncthreads();
    for (int wv = 0; wv < kMaskThreads / kWave; ++wv) {
      bx1 = s_red[0][wv] < bx1 ? s_red[0][wv] : bx1; bx2 = s_red[1][wv] > bx2 ? s_red[1][wv] : bx2;
      by1 = s_red[2][wv] < by1 ? s_red[2][wv] : by1; by2 = s_red[3][wv] > by2 ? s_red[3][wv] : by2;
    }
  }
  unsigned long long* tog = a.bits + (long)row * 2 * a.bit_words;
  unsigned long long* acc = tog + a.bit_words;
  double sums[2] = {0.0, 0.0};  // crop, full
  bool too_large = false;
  for (int pass = 0; pass < 2; ++pass) {
    int H, W;
    double ox, oy;  // subtracted from the polygon coordinates
    if (pass == 0) {  // :40-43, :57,:64
      W = (int)roi.z - (int)roi.x + 1;
      H = (int)roi.w - (int)roi.y + 1;
      ox = (double)roi.x;
      oy = (double)roi.y;
    } else {  // :74-96
      W = (int)bx2 - (int)bx1 + 1;
      H = (int)by2 - (int)by1 + 1;
      ox = bx1;
      oy = by1;
    }
    const long area = (long)H * W;
    if (H <= 0 || W <= 0) continue;  // an empty raster has no pixels
    if (area > (long)a.bit_words * 64) {
      too_large = true;
      continue;
    }
    const int nw = (int)((area + 63) >> 6);
    const int chunk = (nw + kMaskThreads - 1) / kMaskThreads;
    const int w0 = iminr(tid * chunk, nw), w1 = iminr(w0 + chunk, nw);
    for (int wd = w0; wd < w1; ++wd) { tog[wd] = 0ull; acc[wd] = 0ull; }
    int offset = 2 + n_seg;
    for (int sgi = 0; sgi < n_seg; ++sgi) {
      const int cur_len = (int)poly[sgi + 2];
      int k = cur_len / 2;
      if (k > kMaskMaxVerts) k = kMaskMaxVerts;
      __syncthreads();  // vx / vy / start of the previous segment are no longer read
      for (int t = tid; t < k; t += kMaskThreads) {
        const double px = (double)poly[offset + 2 * t], py = (double)poly[offset + 2 * t + 1];
        vx[t] = (int)(5.0 * (px - ox) + .5);  // the mask API's x = image x here (no swap, :57,:64)
        vy[t] = (int)(5.0 * (py - oy) + .5);
      }
      __syncthreads();
      if (tid == 0 && k > 0) { vx[k] = vx[0]; vy[k] = vy[0]; }
      __syncthreads();
      int run = 0;
      for (int base = 0; base < k; base += kMaskThreads) {
        const int e = base + tid;
        int cnt = 0;
        if (e < k) {
          const int dx = abs(vx[e] - vx[e + 1]), dy = abs(vy[e] - vy[e + 1]);
          cnt = (dx > dy ? dx : dy) + 1;
        }
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < kWave; o <<= 1) {
          const int t = __shfl_up(incl, o);
          if ((tid & (kWave - 1)) >= o) incl += t;
        }
        __syncthreads();
        if ((tid & (kWave - 1)) == kWave - 1) wave_sums[tid / kWave] = incl;
        __syncthreads();
        int woff = 0, tot = 0;
        for (int wv = 0; wv < kMaskThreads / kWave; ++wv) {
          if (wv < tid / kWave) woff += wave_sums[wv];
          tot += wave_sums[wv];
        }
        if (e < k) start[e] = run + woff + incl - cnt;
        run += tot;
      }
      if (tid == 0) { start[k] = run; s_total = run; }
      // the zeroed words must be in L2 before another lane's atomic lands on them
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __syncthreads();
      const int m = s_total;
      for (int gidx = 1 + tid; gidx < m; gidx += kMaskThreads) {
        int lo = 0, hi = k - 1;
        while (lo < hi) {
          const int mid = (lo + hi + 1) >> 1;
          if (start[mid] <= gidx) lo = mid; else hi = mid - 1;
        }
        int u1, v1, u0, v0;
        poly_point(vx, vy, lo, gidx - start[lo], &u1, &v1);
        if (gidx - 1 >= start[lo]) poly_point(vx, vy, lo, gidx - 1 - start[lo], &u0, &v0);
        else poly_point(vx, vy, lo - 1, gidx - 1 - start[lo - 1], &u0, &v0);
        if (u1 != u0) {
          double xd = (double)(u1 < u0 ? u1 : u1 - 1);
          xd = (xd + .5) / 5.0 - .5;
          if (floor(xd) != xd || xd < 0 || xd > (double)(W - 1)) continue;
          double yd = (double)(v1 < v0 ? v1 : v0);
          yd = (yd + .5) / 5.0 - .5;
          if (yd < 0) yd = 0; else if (yd > (double)H) yd = (double)H;
          yd = ceil(yd);
          const long pos = (long)xd * H + (long)yd;
          if (pos < area) atomicXor(&tog[pos >> 6], 1ull << (pos & 63));
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      __syncthreads();
      // prefix parity: parity of this lane's words, exclusive xor-scan over the lanes
      int par = 0;
      for (int wd = w0; wd < w1; ++wd)
        par ^= __popcll(__hip_atomic_load(&tog[wd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & 1;
      s_par[tid] = par;
      __syncthreads();
      int carry = 0;
      for (int t = 0; t < tid; ++t) carry ^= s_par[t];
      for (int wd = w0; wd < w1; ++wd) {
        const unsigned long long bitsw = __hip_atomic_load(&tog[wd], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long pp = bitsw;
        pp ^= pp << 1; pp ^= pp << 2; pp ^= pp << 4; pp ^= pp << 8; pp ^= pp << 16; pp ^= pp << 32;
        if (carry) pp = ~pp;
        carry ^= __popcll(bitsw) & 1;
        acc[wd] |= pp;
        tog[wd] = 0ull;
      }
      offset += cur_len;
    }
    // pixels of the union; the bits of the last word past the raster are not pixels
    long cnt = 0;
    for (int wd = w0; wd < w1; ++wd) {
      unsigned long long v = acc[wd];
      if (wd == nw - 1 && (area & 63)) v &= (1ull << (area & 63)) - 1ull;
      cnt += __popcll(v);
    }
    double dc = (double)cnt;
#pragma unroll
    for (int o = kWave / 2; o > 0; o >>= 1) dc += __shfl_xor(dc, o);
    __syncthreads();
    if ((tid & (kWave - 1)) == 0) s_red[0][tid / kWave] = dc;
    __syncthreads();
    double tot = 0;
    for (int wv = 0; wv < kMaskThreads / kWave; ++wv) tot += s_red[0][wv];
    sums[pass] = tot;
    __syncthreads();
  }
  if (tid == 0) {
    double ratio = sums[0] / (sums[1] + 0.0001);
    ratio = ratio > 1e-10 ? ratio : 1e-10;
    a.ratio_out[row] = too_large ? __int_as_float(0x7fc00000) : (float)ratio;
  }
}

static inline size_t align_up(size_t v, size_t al) { return (v + al - 1) / al * al; }

static size_t pt_layout(int B, int N, int M, int S, PtWs* ws, char* base) {
  if (M < 1) M = 1;  // ProposalTarget_v2 substitutes one zero gt / roi for an empty list
  const size_t Ncand = (size_t)N + M;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off = align_up(off + bytes, 256);
    return o;
  };
  const size_t o_cand = take(B * Ncand * 16), o_ov = take(B * Ncand * 4), o_gta = take(B * Ncand * 4);
  const size_t o_gtbox = take((size_t)B * M * 16), o_gtcls = take((size_t)B * M * 4);
  const size_t o_fg = take(B * Ncand * 4), o_neg = take(B * Ncand * 4), o_bg = take(B * Ncand * 4);
  const size_t o_cnt = take((size_t)B * 8 * 4), o_kept = take((size_t)B * (S > 0 ? S : 1) * 4);
  const size_t o_gtsrc = take((size_t)B * M * 4);
  if (ws) {
    ws->cand = reinterpret_cast<float4*>(base + o_cand);
    ws->ov = reinterpret_cast<float*>(base + o_ov);
    ws->gta = reinterpret_cast<int*>(base + o_gta);
    ws->gtbox = reinterpret_cast<float4*>(base + o_gtbox);
    ws->gtcls = reinterpret_cast<float*>(base + o_gtcls);
    ws->fg = reinterpret_cast<int*>(base + o_fg);
    ws->neg = reinterpret_cast<int*>(base + o_neg);
    ws->bg = reinterpret_cast<int*>(base + o_bg);
    ws->counts = reinterpret_cast<int*>(base + o_cnt);
    ws->kept = reinterpret_cast<int*>(base + o_kept);
    ws->gtsrc = reinterpret_cast<int*>(base + o_gtsrc);
  }
  return off;
}

constexpr int kPtMaxRois = 4096;  // S bound for the workspace query (kept list)

}  // namespace sd

using namespace sd;

extern "C" int sd_glibc_srand_host(uint32_t seed, int32_t* state_host) {
  SD_REQUIRE(state_host, "state_host is null");
  // glibc stdlib/random_r.c __srandom_r for TYPE_3 (degree 31, separation 3)
  int32_t r[31];
  if (seed == 0) seed = 1;
  r[0] = (int32_t)seed;
  int32_t word = (int32_t)seed;
  for (int i = 1; i < 31; ++i) {
    const long hi = word / 127773, lo = word % 127773;
    long w = 16807 * lo - 2836 * hi;
    if (w < 0) w += 2147483647;
    word = (int32_t)w;
    r[i] = word;
  }
  int f = 3, b = 0;
  uint32_t* u = reinterpret_cast<uint32_t*>(r);
  for (int i = 0; i < 310; ++i) {
    u[f] += u[b];
    if (++f >= 31) f = 0;
    if (++b >= 31) b = 0;
  }
  for (int i = 0; i < 31; ++i) state_host[i] = r[i];
  state_host[31] = f;
  state_host[32] = b;
  return SD_OK;
}

extern "C" size_t sd_proposal_target_workspace_bytes(int B, int N, int M) {
  if (B <= 0 || N < 0 || M < 0) return 256;
  return pt_layout(B, N, M, kPtMaxRois, nullptr, nullptr) + 256;
}

static int proposal_target_impl(const float* rois, const float* gt_boxes, const float* valid_ranges,
                                int filter_scales, int v2, int N, int M,
                                const sd_proposal_target_param* param_host, int32_t* rng_state,
                                float* roi_output, float* label, float* bbox_target,
                                float* bbox_weight, float* match_gt_iou, int32_t* kept_index,
                                void* workspace, size_t workspace_bytes, void* stream,
                                const float* gt_polys = nullptr, int L = 0, int mask_size = 0,
                                float* mask_target = nullptr, float* mask_ratio = nullptr,
                                int max_raster_pixels = 0) {
  SD_REQUIRE(param_host, "param is null");
  const sd_proposal_target_param& p = *param_host;
  const int B = p.batch_images, S = p.image_rois;
  SD_REQUIRE(B >= 0 && N >= 0 && M >= 0, "negative dimension");
  SD_REQUIRE(S >= 0 && S <= kPtMaxRois, "image_rois=%d outside [0,%d]", S, kPtMaxRois);
  SD_REQUIRE(p.num_classes >= 1, "num_classes must be >= 1");
  SD_REQUIRE(p.fg_fraction >= 0.f && p.fg_fraction <= 1.f, "fg_fraction outside [0,1]");
  for (int k = 0; k < 4; ++k) SD_REQUIRE(p.bbox_std[k] != 0.f, "bbox_std[%d] is zero", k);
  if (B == 0 || S == 0) return SD_OK;
  SD_REQUIRE(roi_output && label && bbox_target && bbox_weight && match_gt_iou,
             "null output pointer (ProposalTarget requires kWriteTo on every output)");
  SD_REQUIRE((rois || N == 0) && (gt_boxes || M == 0), "null input pointer");
  SD_REQUIRE(rng_state, "rng_state is null");
  SD_REQUIRE((((uintptr_t)rois | (uintptr_t)roi_output) & 15) == 0, "rois must be 16-byte aligned");
  PtArgs a{};
  char* base = reinterpret_cast<char*>(align_up((size_t)(uintptr_t)workspace, 256));
  size_t need = pt_layout(B, N, M, S, &a.ws, base) + (size_t)(base - (char*)workspace);
  if (mask_ratio) {  // two bitmaps of max_raster_pixels bits per foreground row behind the rest
    const int FG = (int)((float)S * p.fg_fraction);
    a.bit_words = (max_raster_pixels + 63) / 64;
    a.bits = reinterpret_cast<unsigned long long*>((char*)workspace + align_up(need, 256));
    need = align_up(need, 256) + (size_t)B * FG * 2 * a.bit_words * sizeof(unsigned long long);
    a.ratio_out = mask_ratio;
  }
  if (!workspace || workspace_bytes < need)
    return fail(SD_ERR_WORKSPACE, "ProposalTarget workspace too small: %zu < %zu bytes",
                workspace_bytes, need);
  a.rois = rois; a.gt = gt_boxes; a.p = p;
  a.valid_ranges = valid_ranges; a.filter_scales = valid_ranges ? filter_scales : 0; a.v2 = v2;
  a.polys = gt_polys; a.mask_out = mask_target; a.L = L; a.mask_size = mask_size;
  a.FG = (int)((float)S * p.fg_fraction);
  a.B = B; a.N = N; a.M = M; a.Mws = M > 0 ? M : 1; a.Ncand = N + a.Mws; a.S = S;
  a.fg_per_image = (int)((float)S * p.fg_fraction);  // static_cast<index_t>(image_rois * fg_fraction)
  a.rng = rng_state;
  a.roi_out = roi_output; a.label = label; a.bbox_target = bbox_target;
  a.bbox_weight = bbox_weight; a.iou_out = match_gt_iou; a.kept_index = kept_index;
  hipStream_t st = (hipStream_t)stream;

  const size_t lds1 = (size_t)(M > 0 ? M : 1) * 5 * sizeof(float) + 32;
  SD_REQUIRE(lds1 <= 64 * 1024, "too many gt boxes per image (M=%d)", M);
  hipLaunchKernelGGL((pt_assign_kernel<1024>), dim3(B), dim3(1024), lds1, st, a);
  SD_LAUNCH_CHECK();
  const size_t lds2 = (size_t)(a.Ncand > 0 ? a.Ncand : 1) * 3 * sizeof(int);
  SD_REQUIRE(lds2 <= 150 * 1024, "too many candidate rois per image (%d)", a.Ncand);
  if (lds2 > 64 * 1024)
    SD_HIP_CHECK(hipFuncSetAttribute((const void*)pt_sample_kernel<512>,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds2));
  hipLaunchKernelGGL((pt_sample_kernel<512>), dim3(1), dim3(512), lds2, st, a);
  SD_LAUNCH_CHECK();
  hipLaunchKernelGGL(pt_encode_kernel, dim3(B * S), dim3(128), 0, st, a);
  SD_LAUNCH_CHECK();
  if (mask_target && a.FG > 0) {
    const size_t lds3 = (size_t)(3 * (kMaskMaxVerts + 1) + 2 * mask_size * mask_size) * sizeof(int);
    SD_REQUIRE(lds3 <= 64 * 1024, "mask_size=%d too large", mask_size);
    hipLaunchKernelGGL(pt_mask_kernel, dim3(B * a.FG), dim3(kMaskThreads), lds3, st, a);
    SD_LAUNCH_CHECK();
    if (mask_ratio) {
      hipLaunchKernelGGL(pt_mask_ratio_kernel, dim3(B * a.FG), dim3(kMaskThreads), 0, st, a);
      SD_LAUNCH_CHECK();
    }
  }
  return SD_OK;
}

extern "C" int sd_proposal_target(const float* rois, const float* gt_boxes, int N, int M,
                                  const sd_proposal_target_param* param_host, int32_t* rng_state,
                                  float* roi_output, float* label, float* bbox_target,
                                  float* bbox_weight, float* match_gt_iou, int32_t* kept_index,
                                  void* workspace, size_t workspace_bytes, void* stream) {
  return proposal_target_impl(rois, gt_boxes, nullptr, 0, 0, N, M, param_host, rng_state, roi_output,
                              label, bbox_target, bbox_weight, match_gt_iou, kept_index, workspace,
                              workspace_bytes, stream);
}

extern "C" size_t sd_proposal_mask_target_ratio_workspace_bytes(int B, int N, int M, int image_rois,
                                                                float fg_fraction,
                                                                int max_raster_pixels) {
  if (B <= 0 || N < 0 || M < 0 || image_rois < 0 || max_raster_pixels <= 0) return 256;
  const int FG = (int)((float)image_rois * fg_fraction);
  const size_t words = ((size_t)max_raster_pixels + 63) / 64;
  return align_up(pt_layout(B, N, M, kPtMaxRois, nullptr, nullptr) + 256, 256) + 256 +
         (size_t)B * (FG > 0 ? FG : 0) * 2 * words * sizeof(unsigned long long);
}

extern "C" int sd_proposal_mask_target_ratio(const float* rois, const float* gt_boxes,
                                             const float* gt_polys, const float* valid_ranges,
                                             int filter_scales, int N, int M, int L, int mask_size,
                                             const sd_proposal_target_param* param_host,
                                             int32_t* rng_state, float* roi_output, float* label,
                                             float* bbox_target, float* bbox_weight,
                                             float* match_gt_iou, float* mask_target,
                                             float* mask_ratio, int max_raster_pixels,
                                             int32_t* kept_index, void* workspace,
                                             size_t workspace_bytes, void* stream) {
  SD_REQUIRE(param_host, "param is null");
  SD_REQUIRE(param_host->image_rois >= 0,
             "ProposalMaskTarget: image_rois=-1 is undefined in the reference (negative tensor shape)");
  SD_REQUIRE(mask_size > 0 && L >= 2, "bad mask_size / polygon length");
  SD_REQUIRE((gt_polys && mask_target && mask_ratio) || param_host->batch_images == 0,
             "gt_polys / mask_target / mask_ratio is null");
  SD_REQUIRE(max_raster_pixels > 0, "max_raster_pixels must be positive");
  SD_REQUIRE(!filter_scales || valid_ranges, "filter_scales needs valid_ranges (num_args = 4)");
  return proposal_target_impl(rois, gt_boxes, valid_ranges, filter_scales, 1, N, M, param_host,
                              rng_state, roi_output, label, bbox_target, bbox_weight, match_gt_iou,
                              kept_index, workspace, workspace_bytes, stream, gt_polys, L, mask_size,
                              mask_target, mask_ratio, max_raster_pixels);
}

extern "C" int sd_proposal_target_v2(const float* rois, const float* gt_boxes,
                                     const float* valid_ranges, int filter_scales, int N, int M,
                                     const sd_proposal_target_param* param_host,
                                     int32_t* rng_state, float* roi_output, float* label,
                                     float* bbox_target, float* bbox_weight, float* match_gt_iou,
                                     int32_t* kept_index, void* workspace, size_t workspace_bytes,
                                     void* stream) {
  SD_REQUIRE(param_host, "param is null");
  SD_REQUIRE(valid_ranges || param_host->batch_images == 0, "valid_ranges is null");
  // image_rois == -1 ("keep every roi") makes the reference allocate (B, -1, .) host tensors
  // (proposal_target_v2-inl.h:209-213): undefined there, rejected here
  SD_REQUIRE(param_host->image_rois >= 0,
             "ProposalTarget_v2: image_rois=-1 is undefined in the reference (negative tensor shape)");
  return proposal_target_impl(rois, gt_boxes, valid_ranges, filter_scales, 1, N, M, param_host,
                              rng_state, roi_output, label, bbox_target, bbox_weight, match_gt_iou,
                              kept_index, workspace, workspace_bytes, stream);
}

extern "C" int sd_proposal_mask_target(const float* rois, const float* gt_boxes,
                                       const float* gt_polys, const float* valid_ranges,
                                       int filter_scales, int N, int M, int L, int mask_size,
                                       const sd_proposal_target_param* param_host,
                                       int32_t* rng_state, float* roi_output, float* label,
                                       float* bbox_target, float* bbox_weight, float* match_gt_iou,
                                       float* mask_target, int32_t* kept_index, void* workspace,
                                       size_t workspace_bytes, void* stream) {
  SD_REQUIRE(param_host, "param is null");
  SD_REQUIRE(param_host->image_rois >= 0,
             "ProposalMaskTarget: image_rois=-1 is undefined in the reference (negative tensor shape)");
  SD_REQUIRE(mask_size > 0 && L >= 2, "bad mask_size / polygon length");
  SD_REQUIRE((gt_polys && mask_target) || param_host->batch_images == 0, "gt_polys / mask_target is null");
  SD_REQUIRE(!filter_scales || valid_ranges, "filter_scales needs valid_ranges (num_args = 4)");
  return proposal_target_impl(rois, gt_boxes, valid_ranges, filter_scales, 1, N, M, param_host,
                              rng_state, roi_output, label, bbox_target, bbox_weight, match_gt_iou,
                              kept_index, workspace, workspace_bytes, stream, gt_polys, L, mask_size,
                              mask_target);
}
