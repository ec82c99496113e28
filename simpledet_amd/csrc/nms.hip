// _contrib_NMS for gfx950: sort -> suppression bit matrix -> greedy scan, all on the device.
//   reference: operator_cxx/contrib/nms.cu:92-100 (devIoU), :102-147 (nms_kernel), :149-202 (_nms:
//              mask D2H + host scan + keep H2D), :207-233 (PrepareOutput), :249-365 (Forward:
//              thrust::stable_sort_by_key descending, pre/post top-n); the same kernel with >= is
//              embedded in proposal_v3.cu:271-381.
// MI355X design
//   1. nms_sort_kernel    one workgroup per image: 64-bit keys (score descending | row ascending
//                         == a stable descending sort) bitonic-sorted in LDS (up to 16384 keys =
//                         128 KB of the CU's 160 KB), top `pre` boxes gathered into a float4 array.
//   2. nms_mask_kernel    one wave per 64x64 tile of the UPPER triangle only (the scan never
//                         reads the lower one), stored TRANSPOSED (per column box: which earlier
//                         boxes suppress it): lane = row box in registers, the 64 column boxes
//                         are broadcast through SGPRs (v_readlane) and a column's 64-bit word is
//                         exactly the wave's v_cmp result (ballot) -- no shifts, no LDS.
//   3. nms_scan_kernel    one wave per image replaces the reference's host loop, with no barrier:
//                         lane = box of the current 64-box block, "already suppressed" is an OR of
//                         (T word & keep word) along the box's own row of T (independent loads),
//                         the diagonal word is resolved in rounds of two ballots: every live box
//                         without a live suppressor is kept at once (rounds = longest chain).
//                         Kept boxes are written straight to out/score; the tail is zero padded.
// No host round trip, no device synchronisation, one stream.
#include "common.h"
#include "../../include/simpledet_ops.h"
#include <math.h>

namespace sd {

constexpr int kMaxSortKeys = 16384;

// Sort key of a score: smaller key = better score.  -0.0 and +0.0 get the SAME key (the reference's
// thrust::stable_sort_by_key(greater<float>) and MXNet's SortByKey compare them equal and keep
// their input order; the index in the low half of the 64-bit key does the same here).  NaN scores,
// which the reference's comparator leaves in an unspecified place, are ordered by their bits:
// positive NaNs before +inf, negative NaNs after -inf.
__device__ __forceinline__ unsigned ordered_desc_bits(float f) {
  unsigned u = __float_as_uint(f);
  if (u == 0x80000000u) u = 0u;                     // -0.0 == +0.0
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending-order-preserving map
  return ~u;                                        // descending
}

// ascending bitonic sort of P2 (power of two) 64-bit keys in LDS by the whole workgroup.
// Stages with a pair distance j <= 64 only exchange inside aligned 128-key blocks, so a wave that
// owns such a block runs them back to back with no workgroup barrier (the LDS accesses of one wave
// are ordered): 2048 keys need 15 barriers instead of 66.
__device__ __forceinline__ void bitonic_stage(unsigned long long* keys, int t, int j, int k) {
  // pair (lo, lo + j) of the bitonic network; direction from bit k of lo
  const int lo = ((t & ~(j - 1)) << 1) | (t & (j - 1));
  const int hi = lo + j;
  const unsigned long long x = keys[lo], y = keys[hi];
  const bool up = (lo & k) == 0;
  if ((x > y) == up) {
    keys[lo] = y;
    keys[hi] = x;
  }
}

__device__ __forceinline__ void bitonic_sort_lds(unsigned long long* keys, int P2, int tid, int T) {
  if (P2 < 128 || (T & (kWave - 1))) {  // tiny inputs: every stage with a barrier
    for (int k = 2; k <= P2; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int t = tid; t < P2 / 2; t += T) bitonic_stage(keys, t, j, k);
        __syncthreads();
      }
    return;
  }
  const int lane = tid & (kWave - 1), wave = tid / kWave, nwaves = T / kWave;
  // phases 2 .. 128: every aligned 128-key block is sorted by one wave on its own
  for (int b = wave; b < P2 / 128; b += nwaves)
    for (int k = 2; k <= 128; k <<= 1)
      for (int j = k >> 1; j > 0; j >>= 1) {
        bitonic_stage(keys, b * kWave + lane, j, k);
        wave_lds_sync();
      }
  __syncthreads();
  for (int k = 256; k <= P2; k <<= 1) {
    for (int j = k >> 1; j >= 128; j >>= 1) {  // pairs across blocks: whole workgroup + barrier
      for (int t = tid; t < P2 / 2; t += T) bitonic_stage(keys, t, j, k);
      __syncthreads();
    }
    for (int b = wave; b < P2 / 128; b += nwaves)  // j = 64 .. 1 inside the blocks
      for (int j = 64; j > 0; j >>= 1) {
        bitonic_stage(keys, b * kWave + lane, j, k);
        wave_lds_sync();
      }
    __syncthreads();
  }
}

struct NmsWs {
  int* order;          // (B, pre) original row of the i-th sorted box
  float4* boxes;       // (B, pre)
  float* score;        // (B, pre)
  unsigned long long* mask;  // (B, pre, nb)
};

struct SortArgs {
  const float* dets;
  NmsWs ws;
  int N, pre, P2, already_sorted;
};

__global__ __launch_bounds__(1024) void nms_sort_kernel(SortArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
  const int img = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
  const float* d = a.dets + (long)img * a.N * 5;
  if (!a.already_sorted) {
    for (int i = tid; i < a.P2; i += T) {
      unsigned long long k = ~0ull;
      if (i < a.N) k = ((unsigned long long)ordered_desc_bits(d[(long)i * 5 + 4]) << 32) | (unsigned)i;
      keys[i] = k;
    }
    __syncthreads();
    bitonic_sort_lds(keys, a.P2, tid, T);
  }
  for (int i = tid; i < a.pre; i += T) {
    const int src = a.already_sorted ? i : (int)(unsigned)(keys[i] & 0xffffffffu);
    const float* p = d + (long)src * 5;
    const long o = (long)img * a.pre + i;
    a.ws.order[o] = src;
    a.ws.boxes[o] = make_float4(p[0], p[1], p[2], p[3]);
    a.ws.score[o] = p[4];
  }
}

struct MaskArgs {
  NmsWs ws;
  int pre, nb, npairs;
  float thr;
  int ge;
};

// nms.cu:92-100 with a = the row ("cur") box, b = the column box
__device__ __forceinline__ float dev_iou(float a0, float a1, float a2, float a3, float Sa, float b0,
                                         float b1, float b2, float b3, float Sb) {
  const float left = fmaxr(a0, b0), right = fminr(a2, b2);
  const float top = fmaxr(a1, b1), bottom = fminr(a3, b3);
  const float width = fmaxr(right - left + 1.f, 0.f), height = fmaxr(bottom - top + 1.f, 0.f);
  const float interS = width * height;
  return interS / (Sa + Sb - interS);
}

// Transposed suppression matrix, word-major: T[rb][col] = bit r set  <=>  box (rb*64 + r) suppresses
// box col (IoU over the threshold, r ranked before col).  Word-major so that the 64 lanes of the
// scan (consecutive boxes) read one contiguous 512-byte run per word instead of 64 rows.  Lanes hold the 64 row boxes of block rb, the 64
// column boxes of block cb are broadcast through SGPRs; the word of one column IS the wave's
// v_cmp result (ballot).  devIoU is symmetric in its arguments bit for bit (max/min and the float
// additions commute), so this is the reference's mask read column-wise.
__global__ __launch_bounds__(256) void nms_mask_kernel(MaskArgs a) {
  const int lane = threadIdx.x & (kWave - 1);
  const int pair = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
  const int img = blockIdx.y;
  if (pair >= a.npairs) return;
  // pair -> (rb, cb) with cb >= rb, row-major over the upper triangle
  int rb = 0, rem = pair;
  while (rem >= a.nb - rb) {
    rem -= a.nb - rb;
    ++rb;
  }
  const int cb = rb + rem;
  const float4* boxes = a.ws.boxes + (long)img * a.pre;
  const int row = rb * kWave + lane, col = cb * kWave + lane;
  float4 rbx = make_float4(0.f, 0.f, 0.f, 0.f), cbx = rbx;
  if (row < a.pre) rbx = boxes[row];
  if (col < a.pre) cbx = boxes[col];
  const float Sr = (rbx.z - rbx.x + 1.f) * (rbx.w - rbx.y + 1.f);
  const float Sc = (cbx.z - cbx.x + 1.f) * (cbx.w - cbx.y + 1.f);
  const unsigned long long rowvalid = __ballot(row < a.pre);
  unsigned long long word = 0;
#pragma unroll 8
  for (int c = 0; c < kWave; ++c) {
    const float b0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cbx.x), c));
    const float b1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cbx.y), c));
    const float b2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cbx.z), c));
    const float b3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cbx.w), c));
    const float Sb = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Sc), c));
    // nms.cu:92-100 with a = the row ("cur") box, b = the column box
    const float iou = dev_iou(rbx.x, rbx.y, rbx.z, rbx.w, Sr, b0, b1, b2, b3, Sb);
    unsigned long long m = __ballot(a.ge ? iou >= a.thr : iou > a.thr) & rowvalid;
    if (rb == cb) m &= (1ull << c) - 1;  // only rows ranked before the column (start = tid + 1)
    if (lane == c) word = m;
  }
  if (col < a.pre) a.ws.mask[((long)img * a.nb + rb) * a.pre + col] = word;
}

struct ScanArgs {
  NmsWs ws;
  float* out;
  float* score;
  int* keep_index;
  int pre, post, nb;
  int pad_cyclic;  // 0: zero padding (nms.cu:224-230); 1: repeat the kept boxes (proposal_v3.cu
                   // PrepareOutput with is_train)
};

// Greedy scan, one wave per image, no barrier and no host: lane = box of the current 64-box
// block.  suppressed(box) = OR over earlier blocks w of (T[box][w] & keep[w]) -- independent
// 8-byte loads along the box's own row of T -- then the block's diagonal word is resolved with
// scalar ops: every kept k knocks out the boxes whose diagonal word has bit k (one ballot).
__global__ __launch_bounds__(64) void nms_scan_kernel(ScanArgs a) {
  __shared__ unsigned long long kw[kMaxSortKeys / 64];
  const int img = blockIdx.x, lane = threadIdx.x;
  const unsigned long long* T = a.ws.mask + (long)img * a.pre * a.nb;
  const float4* boxes = a.ws.boxes + (long)img * a.pre;
  const float* sscore = a.ws.score + (long)img * a.pre;
  const int* order = a.ws.order + (long)img * a.pre;
  float* out = a.out + (long)img * a.post * 4;
  float* score = a.score + (long)img * a.post;
  int* keep_index = a.keep_index ? a.keep_index + (long)img * a.post : nullptr;
  int nkeep = 0;
  // greedy resolve of one 64-box block given "suppressed by earlier blocks" (acc) and the
  // block's diagonal word; writes the kept boxes, returns the keep mask
  auto resolve = [&](int rb, bool valid, unsigned long long acc, unsigned long long diag,
                     float4 bx, float sc, int ord) {
    // diag (lane j) = the earlier boxes of this block that would suppress box j.  A live box none
    // of whose potential suppressors is still live is kept -- all such boxes at once -- and
    // whatever they suppress leaves the live set: the number of rounds is the longest suppression
    // chain inside the block (1-3 in practice), not the number of kept boxes.
    unsigned long long cur = __ballot(acc != 0 || !valid);  // resolved: suppressed or kept
    unsigned long long keepmask = 0;
    unsigned long long live = ~cur;
    while (live) {
      const unsigned long long fre = __ballot(((live >> lane) & 1ull) && (diag & live) == 0);
      keepmask |= fre;
      const unsigned long long sup = __ballot((diag & fre) != 0);
      cur |= fre | sup;
      live = ~cur;
    }
    if (a.nb > 32 && lane == 0) kw[rb] = keepmask;  // LDS copy only for the generic path
    // PrepareOutput (nms.cu:207-233) for the boxes kept in this block
    if (keepmask & (1ull << lane)) {
      const int rank = nkeep + __popcll(keepmask & ((1ull << lane) - 1));
      if (rank < a.post) {
        reinterpret_cast<float4*>(out)[rank] = bx;
        score[rank] = sc;
        if (keep_index) keep_index[rank] = ord;
      }
    }
    nkeep += __popcll(keepmask);
    return keepmask;
  };
  // everything a block needs from global memory (its row of T, its diagonal row word, its box)
  // is fetched one block ahead so that no load latency sits on the block-to-block chain
  struct Pre {
    float4 bx;
    float sc;
    int ord;
  };
  auto fetch = [&](int c, bool valid) {
    Pre p{make_float4(0.f, 0.f, 0.f, 0.f), 0.f, -1};
    if (valid) {
      p.bx = boxes[c];
      p.sc = sscore[c];
      p.ord = order[c];
    }
    return p;
  };
  constexpr int NBF = 32;  // fast path: the next block's words of T are fetched during this block's resolve
  if (a.nb <= NBF) {
    // The whole scan is ONE wave's instruction stream, so the per-block instruction count is the
    // run time: the keep words of the earlier blocks live in SGPRs (no LDS round trip, used as
    // scalar operands), and every loop over the words is cut off in groups of 8 by a wave-uniform
    // branch (only words <= rb matter).
    unsigned long long wa[NBF + 1], wb[NBF + 1];  // words 0..rb of two consecutive blocks (ping-pong)
    unsigned long long kwr[NBF];                   // keep masks of the resolved blocks (wave-uniform)
#pragma unroll
    for (int w = 0; w <= NBF; ++w) wa[w] = wb[w] = 0;
#pragma unroll
    for (int w = 0; w < NBF; ++w) kwr[w] = 0;
    if (lane < a.pre) wa[0] = T[lane];
    Pre pa = fetch(lane, lane < a.pre), pb = pa;
    // one block: cw / cp = this block's words and box (fetched during the previous block), nw / np
    // receive the next block's
    auto block = [&](int rb, unsigned long long (&cw)[NBF + 1], unsigned long long (&nw)[NBF + 1],
                     const Pre& cp, Pre& np) {
      const int c = rb * kWave + lane;
      const bool valid = c < a.pre;
      // issue the loads of block rb + 1 now; they complete while this block is resolved
      const int cn = c + kWave;
      const bool nvalid = rb + 1 < a.nb && cn < a.pre;
      const unsigned long long* Tn = T + (nvalid ? cn : 0);
#pragma unroll
      for (int g = 0; g <= NBF; g += 8) {
        if (g <= rb + 1) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int w = g + e;
            if (w <= NBF) nw[w] = (nvalid && w <= rb + 1) ? Tn[(long)w * a.pre] : 0ull;
          }
        }
      }
      np = fetch(cn, nvalid);
      unsigned long long acc = 0, diag = 0;
#pragma unroll
      for (int g = 0; g <= NBF; g += 8) {
        if (g <= rb) {
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int w = g + e;
            if (w < NBF && w < rb) acc |= cw[w] & kwr[w];
            if (w <= NBF && w == rb) diag = cw[w];
          }
        }
      }
      const unsigned long long km = resolve(rb, valid, acc, diag, cp.bx, cp.sc, cp.ord);
#pragma unroll
      for (int w = 0; w < NBF; ++w)
        if (w == rb) kwr[w] = km;
    };
    for (int rb = 0; rb < a.nb; rb += 2) {
      if (nkeep >= a.post) break;  // only the first `post` kept boxes are ever output
      block(rb, wa, wb, pa, pb);
      if (rb + 1 >= a.nb || nkeep >= a.post) break;
      block(rb + 1, wb, wa, pb, pa);
    }
  } else {
    for (int rb = 0; rb < a.nb; ++rb) {
      if (nkeep >= a.post) break;
      const int c = rb * kWave + lane;
      const bool valid = c < a.pre;
      const unsigned long long* Tc = T + (valid ? c : 0);
      unsigned long long acc = 0;
      for (int w0 = 0; w0 < rb; w0 += 16) {
        unsigned long long t[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) t[e] = (w0 + e < rb) ? Tc[(long)(w0 + e) * a.pre] : 0ull;
#pragma unroll
        for (int e = 0; e < 16; ++e)
          if (w0 + e < rb) acc |= t[e] & kw[w0 + e];
      }
      const Pre p = fetch(c, valid);
      (void)resolve(rb, valid, acc, valid ? Tc[(long)rb * a.pre] : 0ull, p.bx, p.sc, p.ord);
    }
  }
  if (nkeep > a.post) nkeep = a.post;
  __threadfence_block();  // the kept rows written above are read back below (same wave)
  for (int i = nkeep + lane; i < a.post; i += kWave) {
    if (a.pad_cyclic && nkeep > 0) {
      const int src = i % nkeep;
      reinterpret_cast<float4*>(out)[i] = reinterpret_cast<const float4*>(out)[src];
      score[i] = score[src];
      if (keep_index) keep_index[i] = keep_index[src];
    } else {
      reinterpret_cast<float4*>(out)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
      score[i] = 0.f;
      if (keep_index) keep_index[i] = -1;
    }
  }
}


// ------------------------------------------------------------------------------------------------
// _contrib_Proposal_v3: anchors -> decode -> top-k -> min-size filter -> NMS (>=) -> padded output
//   reference: operator_cxx/contrib/proposal_v3.cu:64-235 (grid, decode, filter kernels), :428-638
//   (Forward: per image thrust sort of ALL anchors, 3 D2H/H2D copies and a host NMS scan).
// Here: one decode launch for the batch, one workgroup per image that SELECTS the top
// pre_nms_top_n scores with an 8-bit radix select over the score keys (4 passes over L2-resident
// keys, no full sort of the 10^5 anchors), bitonic-sorts only the selected <= 16384 keys in LDS,
// applies the min-size filter, then the NMS mask / scan kernels above.  im_info stays on the device.
// ------------------------------------------------------------------------------------------------
constexpr int kMaxAnchors = 64;

struct PropArgs {
  const float* cls_prob;   // (B, 2A, H, W)
  const float* bbox_pred;  // (B, 4A, H, W)
  const float* im_info;    // (B, 3)
  float4* boxes_all;       // (B, count)
  float* score_all;        // (B, count)
  NmsWs ws;
  float anchors[kMaxAnchors * 4];
  int A, H, W, stride, count, pre, P2;
  int iou_loss;  // IoUPredKernel instead of BBoxPredKernel
  float min_size;
  // multi-workgroup top-k (large levels): per image 4 x 256 digit counters, a candidate counter and
  // P2 candidate keys in global memory; G workgroups per image
  int* ghist;
  int* gncand;
  unsigned long long* gcand;
  int G;
};

__global__ __launch_bounds__(256) void proposal_decode_kernel(PropArgs a) {
  const int img = blockIdx.y;
  const int index = blockIdx.x * blockDim.x + threadIdx.x;
  if (index >= a.count) return;
  const int an = index % a.A, w = (index / a.A) % a.W, h = index / a.A / a.W;
  const float im_height = a.im_info[img * 3 + 0], im_width = a.im_info[img * 3 + 1];
  const long plane = (long)a.H * a.W;
  const float* fg = a.cls_prob + ((long)img * 2 * a.A + a.A) * plane;
  const float* deltas = a.bbox_pred + (long)img * 4 * a.A * plane;
  const long hw = (long)h * a.W + w;
  // ProposalGridKernel (proposal_v3.cu:64-85)
  const float x1 = a.anchors[an * 4 + 0] + (float)(w * a.stride);
  const float y1 = a.anchors[an * 4 + 1] + (float)(h * a.stride);
  const float x2 = a.anchors[an * 4 + 2] + (float)(w * a.stride);
  const float y2 = a.anchors[an * 4 + 3] + (float)(h * a.stride);
  // BBoxPredKernel (:92-155)
  const float width = x2 - x1 + 1.0f, height = y2 - y1 + 1.0f;
  const float ctr_x = x1 + 0.5f * width, ctr_y = y1 + 0.5f * height;
  const float dx = deltas[(long)(an * 4) * plane + hw];
  const float dy = deltas[(long)(an * 4 + 1) * plane + hw];
  float dw = deltas[(long)(an * 4 + 2) * plane + hw];
  float dh = deltas[(long)(an * 4 + 3) * plane + hw];
  dw = (float)((double)dw < 4.135166556742356 ? (double)dw : 4.135166556742356);
  dh = (float)((double)dh < 4.135166556742356 ? (double)dh : 4.135166556742356);
  const float pred_ctr_x = dx * width + ctr_x, pred_ctr_y = dy * height + ctr_y;
  const float pred_w = (float)exp((double)dw) * width, pred_h = (float)exp((double)dh) * height;
  float px1 = pred_ctr_x - 0.5f * pred_w, py1 = pred_ctr_y - 0.5f * pred_h;
  float px2 = pred_ctr_x + 0.5f * pred_w - 1.0f, py2 = pred_ctr_y + 0.5f * pred_h - 1.0f;
  if (a.iou_loss) {  // IoUPredKernel (:163-205): the four deltas move the anchor's corners
    px1 = x1 + dx;
    py1 = y1 + dy;
    px2 = x2 + deltas[(long)(an * 4 + 2) * plane + hw];
    py2 = y2 + deltas[(long)(an * 4 + 3) * plane + hw];
  }
  float4 o;
  o.x = fmaxr(fminr(px1, im_width - 1.0f), 0.0f);
  o.y = fmaxr(fminr(py1, im_height - 1.0f), 0.0f);
  o.z = fmaxr(fminr(px2, im_width - 1.0f), 0.0f);
  o.w = fmaxr(fminr(py2, im_height - 1.0f), 0.0f);
  a.boxes_all[(long)img * a.count + index] = o;
  float sc = fg[(long)an * plane + hw];
  // IoUPredKernel only (:201-203; commented out in BBoxPredKernel): anchors past the unpadded image
  if (a.iou_loss && (h >= (int)(im_height / (float)a.stride) || w >= (int)(im_width / (float)a.stride)))
    sc = -1.0f;
  a.score_all[(long)img * a.count + index] = sc;
}

// one 8-bit digit of a radix select among the elements whose key matches `prefix` under `mask`:
// returns the digit where the running count reaches `want` (1-based rank inside the matching set)
// and updates below (elements before that digit) -- all threads get the same values
template <typename KeyFn>
__device__ __forceinline__ int radix_digit(int count, int shift, unsigned mask, unsigned prefix,
                                           int want, int* hist, int* below, int* bucket,
                                           KeyFn key_of) {
  const int tid = threadIdx.x, T = blockDim.x;
  for (int i = tid; i < 256; i += T) hist[i] = 0;
  __syncthreads();
  // RPN scores share a handful of exponent bytes, so plain per-lane LDS atomics would serialise on
  // a few counters: lanes with the same digit are merged (ballot) into one atomic per wave
  constexpr int UN = 8;  // keys fetched per thread before any of them is counted (latency)
  for (int i0 = 0; i0 < count; i0 += UN * T) {
    unsigned kk[UN];
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int i = i0 + u * T + tid;
      kk[u] = i < count ? key_of(i) : ~prefix;  // ~prefix never matches under a non-empty mask
    }
#pragma unroll
    for (int u = 0; u < UN; ++u) {
      const int i = i0 + u * T + tid;
      int dg = -1;
      if (i < count && (kk[u] & mask) == prefix) dg = (int)((kk[u] >> shift) & 255);
      unsigned long long todo = __ballot(dg >= 0);
      for (int round = 0; round < 4 && todo; ++round) {  // popular digits first, merged
        const int leader = __ffsll((long long)todo) - 1;
        const int d0 = __builtin_amdgcn_readlane(dg, leader);
        const unsigned long long same = __ballot(dg == d0);
        if ((int)(threadIdx.x & 63) == leader) atomicAdd(&hist[d0], __popcll(same));
        if (dg == d0) dg = -1;
        todo &= ~same;
      }
      if (dg >= 0) atomicAdd(&hist[dg], 1);  // whatever is left is spread over many counters
    }
  }
  __syncthreads();
  // wave 0 scans the 256 counters (4 per lane + a wave prefix sum) and publishes the digit where
  // the running count reaches `want`
  if (threadIdx.x < kWave) {
    const int lane = threadIdx.x;
    const int c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2],
              c3 = hist[4 * lane + 3];
    const int tot = c0 + c1 + c2 + c3;
    int incl = tot;
#pragma unroll
    for (int o = 1; o < kWave; o <<= 1) {
      const int t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    const int excl = incl - tot;
    if (excl < want && want <= incl) {  // exactly one lane
      int run = excl, d = 4 * lane, sz = c0;
      if (run + c0 < want) { run += c0; d = 4 * lane + 1; sz = c1;
        if (run + c1 < want) { run += c1; d = 4 * lane + 2; sz = c2;
          if (run + c2 < want) { run += c2; d = 4 * lane + 3; sz = c3; } } }
      hist[256] = d;
      hist[257] = run;
      hist[258] = sz;
    }
  }
  __syncthreads();
  const int digit = hist[256], lower = hist[257], size = hist[258];
  *below = lower;
  *bucket = size;
  __syncthreads();
  return digit;
}

// gather of the sorted rows + FilterBoxKernel (proposal_v3.cu:211-235)
__device__ __forceinline__ void proposal_gather_filter(const PropArgs& a, int img,
                                                       const unsigned long long* keys, int tid,
                                                       int T) {
  const float* sc = a.score_all + (long)img * a.count;
  const float4* bx = a.boxes_all + (long)img * a.count;
  const float im_h = a.im_info[img * 3 + 0], im_w = a.im_info[img * 3 + 1];
  const float scale = a.im_info[img * 3 + 2];
  for (int i = tid; i < a.pre; i += T) {
    const int src = (int)(unsigned)(keys[i] & 0xffffffffu);
    float4 d = bx[src];
    float s = sc[src];
    const float ws_orig_scale = (d.z - d.x) / scale + 1.0f;
    const float hs_orig_scale = (d.w - d.y) / scale + 1.0f;
    const float min_size_max = fmaxr(a.min_size, 1.0f);
    const float ws = d.z - d.x + 1.0f, hs = d.w - d.y + 1.0f;
    const float x_ctr = d.x + ws / 2.0f, y_ctr = d.y + hs / 2.0f;
    if (ws_orig_scale < min_size_max || hs_orig_scale < min_size_max || x_ctr >= im_w ||
        y_ctr >= im_h) {
      d.x -= min_size_max / 2;
      d.y -= min_size_max / 2;
      d.z += min_size_max / 2;
      d.w += min_size_max / 2;
      s = -1.0f;
    }
    const long o = (long)img * a.pre + i;
    a.ws.order[o] = src;
    a.ws.boxes[o] = d;
    a.ws.score[o] = s;
  }
}

// Stable top-`pre` of `count` scores by one workgroup: 8-bit radix select of the pre-th best score
// key, a second select on the row index when the ties at that key are only partly taken, unordered
// compaction of the selected rows into composite (key, row) words and an LDS sort of the P2 >= pre
// words.  keys[0 .. pre) end up in the order of a stable descending sort.
template <int STRIDE = 1>  // score of row i at sc[i * STRIDE]
__device__ __forceinline__ void select_sort_topk(const float* __restrict__ sc, int count, int pre,
                                                 int P2, unsigned long long* keys, int* hist,
                                                 int* ncand) {
  const int tid = threadIdx.x, T = blockDim.x;
  auto skey = [&](int i) { return ordered_desc_bits(sc[(long)i * STRIDE]); };  // ascending key = best score first
  // ---- the pre-th smallest score key ----
  unsigned prefix = 0, mask = 0;
  int want = pre, last_bucket = 0;
  for (int shift = 24; shift >= 0; shift -= 8) {
    int below;
    const int d = radix_digit(count, shift, mask, prefix, want, hist, &below, &last_bucket, skey);
    want -= below;
    prefix |= (unsigned)d << shift;
    mask |= 255u << shift;
  }
  const unsigned Tkey = prefix;  // `want` of the elements with this key are still needed
  const int n_eq = last_bucket;  // how many elements carry exactly this key
  // ---- ties on the score: the lowest rows win (stable sort); select the want-th smallest row
  //      (skipped when every tied element is taken, the usual case) ----
  unsigned Irow = 0xffffffffu;  // ties with row <= Irow are taken
  if (want < n_eq) {
    unsigned ipre = 0, imask = 0;
    int iwant = want;
    auto ikey = [&](int i) { return skey(i) == Tkey ? (unsigned)i : 0xffffffffu; };
    for (int shift = 24; shift >= 0; shift -= 8) {
      int below, bucket;
      // rows that do not carry Tkey map to 0xffffffff and never match a prefix below 2^24 rows
      const int d = radix_digit(count, shift, imask, ipre, iwant, hist, &below, &bucket, ikey);
      iwant -= below;
      ipre |= (unsigned)d << shift;
      imask |= 255u << shift;
    }
    Irow = ipre;
  }
  // ---- unordered compaction of the selected rows into composite keys, then sort ----
  if (tid == 0) *ncand = 0;
  for (int i = tid; i < P2; i += T) keys[i] = ~0ull;
  __syncthreads();
  for (int i0 = 0; i0 < count; i0 += 8 * T) {
    unsigned kk[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * T + tid;
      kk[u] = i < count ? skey(i) : 0xffffffffu;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * T + tid;
      const unsigned k = kk[u];
      if (i < count && (k < Tkey || (k == Tkey && (unsigned)i <= Irow))) {
        const int pos = atomicAdd(ncand, 1);
        if (pos < P2) keys[pos] = ((unsigned long long)k << 32) | (unsigned)i;
      }
    }
  }
  __syncthreads();
  bitonic_sort_lds(keys, P2, tid, T);
}

__global__ __launch_bounds__(1024) void proposal_topk_kernel(PropArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];  // P2 composite keys
  __shared__ int hist[260];
  __shared__ int ncand;
  const int img = blockIdx.x;
  select_sort_topk(a.score_all + (long)img * a.count, a.count, a.pre, a.P2, keys, hist, &ncand);
  proposal_gather_filter(a, img, keys, threadIdx.x, blockDim.x);
}

// _contrib_NMS with more rows than the LDS sort holds (nms.cu:303 sorts any N): only the first
// pre_nms_top_n rows of the sorted order are ever used (nms.cu:311-313), so they are selected by
// the radix select above (scores at stride 5 inside the (N, 5) rows) and only they are sorted.
__global__ __launch_bounds__(1024) void nms_select_sort_kernel(SortArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];  // P2 >= pre words
  __shared__ int hist[260];
  __shared__ int ncand;
  const int img = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
  const float* d = a.dets + (long)img * a.N * 5;
  select_sort_topk<5>(d + 4, a.N, a.pre, a.P2, keys, hist, &ncand);
  __syncthreads();
  for (int i = tid; i < a.pre; i += T) {
    const int src = (int)(unsigned)(keys[i] & 0xffffffffu);
    const float* p = d + (long)src * 5;
    const long o = (long)img * a.pre + i;
    a.ws.order[o] = src;
    a.ws.boxes[o] = make_float4(p[0], p[1], p[2], p[3]);
    a.ws.score[o] = p[4];
  }
}

// ------------------------------------------------------------------------------------------------
// Multi-workgroup top-k for the large pyramid levels.  The single-workgroup radix select above is
// bound by ONE CU's instruction rate (P2: 200 k scores x 5 passes = 543 us).  Here G workgroups per
// image split the scores: four histogram launches (one 8-bit digit each: local LDS histogram ->
// at most 256 global atomics per workgroup), one compaction launch (selected rows appended to a
// global candidate list with wave-aggregated atomics) and the finishing workgroup (LDS sort of the
// <= P2 candidates, gather, FilterBox).  Every workgroup re-derives the prefix from the global
// counters of the earlier digits (256 loads + a wave scan), so nothing is read back by the host.
// ------------------------------------------------------------------------------------------------
struct TopkState {
  unsigned prefix, mask;
  int want;  // rank still wanted inside the matching set
  int n_eq;  // size of the selected bucket of the last resolved digit
};

// resolve digits 0 .. npass-1 from the global counters (every thread of the calling workgroup gets
// the same values; hist: LDS scratch of 260 ints)
__device__ __forceinline__ TopkState topk_resolve(const int* __restrict__ gh, int npass, int pre,
                                                  int* hist) {
  TopkState s{0u, 0u, pre, 0};
  for (int p = 0; p < npass; ++p) {
    const int shift = 24 - 8 * p;
    __syncthreads();
    if (threadIdx.x < kWave) {
      const int lane = threadIdx.x;
      const int* h = gh + p * 256;
      const int c0 = h[4 * lane], c1 = h[4 * lane + 1], c2 = h[4 * lane + 2], c3 = h[4 * lane + 3];
      const int tot = c0 + c1 + c2 + c3;
      int incl = tot;
#pragma unroll
      for (int o = 1; o < kWave; o <<= 1) {
        const int t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
      }
      const int excl = incl - tot;
      if (excl < s.want && s.want <= incl) {  // exactly one lane
        int run = excl, d = 4 * lane, sz = c0;
        if (run + c0 < s.want) { run += c0; d = 4 * lane + 1; sz = c1;
          if (run + c1 < s.want) { run += c1; d = 4 * lane + 2; sz = c2;
            if (run + c2 < s.want) { run += c2; d = 4 * lane + 3; sz = c3; } } }
        hist[256] = d;
        hist[257] = run;
        hist[258] = sz;
      }
    }
    __syncthreads();
    s.want -= hist[257];
    s.prefix |= (unsigned)hist[256] << shift;
    s.mask |= 255u << shift;
    s.n_eq = hist[258];
  }
  __syncthreads();
  return s;
}

// grid (G, B): digit `pass` of the score keys of this workgroup's chunk
__global__ __launch_bounds__(256) void topk_hist_kernel(PropArgs a, int pass) {
  __shared__ int hist[260];
  __shared__ int lh[256];
  const int img = blockIdx.y, tid = threadIdx.x;
  const float* sc = a.score_all + (long)img * a.count;
  int* gh = a.ghist + (long)img * 4 * 256;
  const TopkState s = topk_resolve(gh, pass, a.pre, hist);
  lh[tid] = 0;
  __syncthreads();
  const int shift = 24 - 8 * pass;
  const int chunk = (a.count + a.G - 1) / a.G;
  const int lo = blockIdx.x * chunk, hi = iminr(lo + chunk, a.count);
  for (int i0 = lo; i0 < hi; i0 += 8 * 256) {  // eight loads in flight per lane
    unsigned kk[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 256 + tid;
      kk[u] = i < hi ? ordered_desc_bits(sc[i]) : ~s.prefix;  // never matches a non-empty mask
    }
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (i0 + u * 256 + tid < hi && (kk[u] & s.mask) == s.prefix)
        atomicAdd(&lh[(kk[u] >> shift) & 255], 1);
  }
  __syncthreads();
  if (lh[tid]) atomicAdd(&gh[pass * 256 + tid], lh[tid]);
}

// grid (G, B): rows strictly better than the threshold key, plus the ties when all of them are taken
__global__ __launch_bounds__(256) void topk_compact_kernel(PropArgs a) {
  __shared__ int hist[260];
  const int img = blockIdx.y, tid = threadIdx.x, lane = tid & (kWave - 1);
  const float* sc = a.score_all + (long)img * a.count;
  const TopkState s = topk_resolve(a.ghist + (long)img * 4 * 256, 4, a.pre, hist);
  const unsigned Tkey = s.prefix;
  const bool all_ties = s.want >= s.n_eq;
  unsigned long long* cand = a.gcand + (long)img * a.P2;
  const int chunk = (a.count + a.G - 1) / a.G;
  const int lo = blockIdx.x * chunk, hi = iminr(lo + chunk, a.count);
  __shared__ int wcount[4], wbase;
  const int wave = tid / kWave;
  for (int i0 = lo; i0 < hi; i0 += 8 * 256) {  // eight loads in flight per lane
    unsigned kk[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 256 + tid;
      kk[u] = i < hi ? ordered_desc_bits(sc[i]) : 0xffffffffu;
    }
    // one global reservation per workgroup and trip (a per-wave atomic on the image's single
    // counter serialises ~4000 atomics at the L2: 30 us for P2)
    unsigned long long bal[8];
    int mine = 0;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 256 + tid;
      const bool take = i < hi && (kk[u] < Tkey || (kk[u] == Tkey && all_ties));
      bal[u] = __ballot(take);
      mine += __popcll(bal[u]);  // wave-uniform
    }
    if (lane == 0) wcount[wave] = mine;
    __syncthreads();
    if (tid == 0) {
      const int tot = wcount[0] + wcount[1] + wcount[2] + wcount[3];
      wbase = tot ? atomicAdd(&a.gncand[img], tot) : 0;
    }
    __syncthreads();
    int base = wbase;
    for (int w = 0; w < wave; ++w) base += wcount[w];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = i0 + u * 256 + tid;
      if ((bal[u] >> lane) & 1ull) {
        const int pos = base + __popcll(bal[u] & ((1ull << lane) - 1));
        if (pos < a.P2) cand[pos] = ((unsigned long long)kk[u] << 32) | (unsigned)i;
      }
      base += __popcll(bal[u]);
    }
    __syncthreads();  // wcount / wbase are reused by the next trip
  }
}

// one workgroup per image: ties that are only partly taken (rare), sort, gather + FilterBox
__global__ __launch_bounds__(1024) void topk_finish_kernel(PropArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];  // P2 composite keys
  __shared__ int hist[260];
  __shared__ int ncand;
  const int img = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
  const float* sc = a.score_all + (long)img * a.count;
  auto skey = [&](int i) { return ordered_desc_bits(sc[i]); };
  const TopkState s = topk_resolve(a.ghist + (long)img * 4 * 256, 4, a.pre, hist);
  const unsigned Tkey = s.prefix;
  const unsigned long long* cand = a.gcand + (long)img * a.P2;
  const int nc = iminr(a.gncand[img], a.P2);
  for (int i = tid; i < a.P2; i += T) keys[i] = i < nc ? cand[i] : ~0ull;
  if (tid == 0) ncand = nc;
  __syncthreads();
  if (s.want < s.n_eq) {
    // only the `want` lowest rows among the ties are taken: select the want-th smallest tied row
    unsigned ipre = 0, imask = 0;
    int iwant = s.want;
    auto ikey = [&](int i) { return skey(i) == Tkey ? (unsigned)i : 0xffffffffu; };
    for (int shift = 24; shift >= 0; shift -= 8) {
      int below, bucket;
      const int d = radix_digit(a.count, shift, imask, ipre, iwant, hist, &below, &bucket, ikey);
      iwant -= below;
      ipre |= (unsigned)d << shift;
      imask |= 255u << shift;
    }
    const unsigned Irow = ipre;
    for (int i = tid; i < a.count; i += T) {
      if (skey(i) == Tkey && (unsigned)i <= Irow) {
        const int pos = atomicAdd(&ncand, 1);
        if (pos < a.P2) keys[pos] = ((unsigned long long)Tkey << 32) | (unsigned)i;
      }
    }
    __syncthreads();
  }
  bitonic_sort_lds(keys, a.P2, tid, T);
  proposal_gather_filter(a, img, keys, tid, T);
}


static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

static void nms_dims(int N, int pre_in, int post_in, int* pre, int* post, int* nb) {
  int p = pre_in > 0 ? pre_in : N;  // nms.cu:274-277
  if (p > N) p = N;
  *pre = p;
  *post = post_in < p ? post_in : p;
  *nb = (p + 63) / 64;
}

static size_t nms_layout(int B, int pre, int nb, NmsWs* ws, char* base) {
  size_t off = 0;
  auto take = [&](size_t bytes) {
    size_t o = off;
    off = align_up(off + bytes, 256);
    return o;
  };
  const size_t o_order = take((size_t)B * pre * sizeof(int));
  const size_t o_boxes = take((size_t)B * pre * sizeof(float4));
  const size_t o_score = take((size_t)B * pre * sizeof(float));
  const size_t o_mask = take((size_t)B * pre * nb * sizeof(unsigned long long));
  if (ws) {
    ws->order = reinterpret_cast<int*>(base + o_order);
    ws->boxes = reinterpret_cast<float4*>(base + o_boxes);
    ws->score = reinterpret_cast<float*>(base + o_score);
    ws->mask = reinterpret_cast<unsigned long long*>(base + o_mask);
  }
  return off;
}

}  // namespace sd

using namespace sd;

extern "C" size_t sd_nms_workspace_bytes(int B, int N, int pre_nms_top_n) {
  if (B <= 0 || N <= 0) return 256;
  int pre, post, nb;
  nms_dims(N, pre_nms_top_n, N, &pre, &post, &nb);
  return nms_layout(B, pre, nb, nullptr, nullptr) + 256;
}

extern "C" int sd_nms(const float* dets, int B, int N, int pre_nms_top_n, int post_nms_top_n,
                      float threshold, int threshold_ge, int already_sorted, float* out,
                      float* score, int32_t* keep_index, void* workspace, size_t workspace_bytes,
                      void* stream) {
  SD_REQUIRE(B >= 0 && N >= 0, "negative dimension");
  SD_REQUIRE(post_nms_top_n >= 0, "post_nms_top_n < 0");
  if (B == 0 || N == 0 || post_nms_top_n == 0) return SD_OK;
  int pre, post, nb;
  nms_dims(N, pre_nms_top_n, post_nms_top_n, &pre, &post, &nb);
  SD_REQUIRE(dets && out && score, "null tensor pointer");
  SD_REQUIRE(((uintptr_t)out & 15) == 0, "out must be 16-byte aligned");
  int P2 = 1;
  while (P2 < N) P2 <<= 1;
  // more rows than one LDS sort holds: select the pre_nms_top_n best first (pre < N), sort those
  const bool select = !already_sorted && P2 > kMaxSortKeys;
  if (select && pre > kMaxSortKeys)
    return fail(SD_ERR_UNSUPPORTED, "NMS: N=%d unsorted rows need pre_nms_top_n <= %d (the in-LDS "
                "sort capacity; got %d), or already_sorted=1 with pre-sorted input", N, kMaxSortKeys, pre);
  SD_REQUIRE(pre <= kMaxSortKeys, "NMS: pre_nms_top_n=%d exceeds %d", pre, kMaxSortKeys);
  SD_REQUIRE(N < (1 << 24), "NMS: N=%d rows per image exceed 2^24", N);
  if (select) {
    P2 = 1;
    while (P2 < pre) P2 <<= 1;
  }
  NmsWs ws;
  char* base = reinterpret_cast<char*>(align_up((size_t)(uintptr_t)workspace, 256));
  const size_t need = nms_layout(B, pre, nb, &ws, base) + (size_t)(base - (char*)workspace);
  if (!workspace || workspace_bytes < need)
    return fail(SD_ERR_WORKSPACE, "NMS workspace too small: %zu < %zu bytes", workspace_bytes, need);
  hipStream_t st = (hipStream_t)stream;

  SortArgs sa{dets, ws, N, pre, P2, already_sorted};
  const size_t lds = already_sorted ? 0 : (size_t)P2 * sizeof(unsigned long long);
  if (lds > 64 * 1024)
    SD_HIP_CHECK(hipFuncSetAttribute((const void*)nms_sort_kernel,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int sort_threads = P2 / 2 >= 1024 ? 1024 : (P2 / 2 >= 64 ? P2 / 2 : 64);
  if (select) {
    if (lds > 64 * 1024)
      SD_HIP_CHECK(hipFuncSetAttribute((const void*)nms_select_sort_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(nms_select_sort_kernel, dim3(B), dim3(1024), lds, st, sa);
  } else {
    hipLaunchKernelGGL(nms_sort_kernel, dim3(B), dim3(sort_threads), lds, st, sa);
  }
  SD_LAUNCH_CHECK();

  MaskArgs ma{ws, pre, nb, nb * (nb + 1) / 2, threshold, threshold_ge};
  hipLaunchKernelGGL(nms_mask_kernel, dim3((ma.npairs + 3) / 4, B), dim3(256), 0, st, ma);
  SD_LAUNCH_CHECK();

  ScanArgs ca{ws, out, score, keep_index, pre, post, nb, 0};
  hipLaunchKernelGGL(nms_scan_kernel, dim3(B), dim3(64), 0, st, ca);
  SD_LAUNCH_CHECK();
  return SD_OK;
}

// proposal_v3-inl.h:279-318 (float math, floor(size / ratio))
static void proposal_v3_anchors(int feature_stride, const float* scales, int ns, const float* ratios,
                                int nr, float* anchors) {
  const float base[4] = {0.0f, 0.0f, (float)(feature_stride - 1.0), (float)(feature_stride - 1.0)};
  int n = 0;
  for (int j = 0; j < nr; ++j)
    for (int k = 0; k < ns; ++k) {
      const float scale = scales[k], ratio = ratios[j];
      const float w = base[2] - base[0] + 1.0f, h = base[3] - base[1] + 1.0f;
      const float x_ctr = (float)(base[0] + 0.5 * (w - 1.0f));
      const float y_ctr = (float)(base[1] + 0.5 * (h - 1.0f));
      const float size_ratios = floorf((w * h) / ratio);
      const float new_w = rintf(sqrtf(size_ratios)) * scale;
      const float new_h = rintf((new_w / scale * ratio)) * scale;
      anchors[n * 4 + 0] = x_ctr - 0.5f * (new_w - 1.0f);
      anchors[n * 4 + 1] = y_ctr - 0.5f * (new_h - 1.0f);
      anchors[n * 4 + 2] = x_ctr + 0.5f * (new_w - 1.0f);
      anchors[n * 4 + 3] = y_ctr + 0.5f * (new_h - 1.0f);
      ++n;
    }
}

static void proposal_dims(int count, int pre_in, int post_in, int is_train, int* pre, int* post) {
  int p = pre_in > 0 ? pre_in : count;  // proposal_v3.cu:467-473
  if (p > count) p = count;
  *pre = p;
  *post = post_in < p ? post_in : p;
  if (!is_train) *post = post_in;
}

extern "C" size_t sd_proposal_v3_workspace_bytes(int B, int A, int H, int W, int pre_nms_top_n) {
  if (B <= 0 || A <= 0 || H <= 0 || W <= 0) return 256;
  const long count = (long)A * H * W;
  int pre, post;
  proposal_dims((int)count, pre_nms_top_n, 1, 1, &pre, &post);
  const int nb = (pre + 63) / 64;
  int P2 = 64;
  while (P2 < pre) P2 <<= 1;
  return nms_layout(B, pre, nb, nullptr, nullptr) + align_up((size_t)B * count * 16, 256) +
         align_up((size_t)B * count * 4, 256) + align_up((size_t)B * (4 * 256 + 1) * 4, 256) +
         align_up((size_t)B * P2 * 8, 256) + 512;
}

static int proposal_v3_impl(const float* cls_prob, const float* bbox_pred, const float* im_info,
                            float* out, float* score, int B, int A, int H, int W,
                            int rpn_pre_nms_top_n, int rpn_post_nms_top_n, float threshold,
                            int rpn_min_size, const float* scales_host, int n_scales,
                            const float* ratios_host, int n_ratios, int feature_stride,
                            int is_train, int iou_loss, void* workspace, size_t workspace_bytes,
                            void* stream) {
  SD_REQUIRE(B >= 0 && A > 0 && H > 0 && W > 0, "bad dimensions");
  SD_REQUIRE(scales_host && ratios_host && n_scales * n_ratios == A,
             "num_anchors (%d) != ratios (%d) x scales (%d)", A, n_ratios, n_scales);
  SD_REQUIRE(A <= kMaxAnchors, "more than %d anchors per location", kMaxAnchors);
  SD_REQUIRE(feature_stride > 0 && rpn_post_nms_top_n >= 0, "bad stride / post_nms_top_n");
  const long count_l = (long)A * H * W;
  SD_REQUIRE(count_l < (1L << 24), "too many anchors per image (%ld)", count_l);
  const int count = (int)count_l;
  int pre, post;
  proposal_dims(count, rpn_pre_nms_top_n, rpn_post_nms_top_n, is_train, &pre, &post);
  if (B == 0 || post == 0) return SD_OK;
  SD_REQUIRE(cls_prob && bbox_pred && im_info && out && score, "null tensor pointer");
  SD_REQUIRE(pre <= kMaxSortKeys, "Proposal: rpn_pre_nms_top_n=%d exceeds %d", pre, kMaxSortKeys);
  SD_REQUIRE(((uintptr_t)out & 15) == 0, "out must be 16-byte aligned");
  const int nb = (pre + 63) / 64;
  PropArgs a{};
  char* base = reinterpret_cast<char*>(align_up((size_t)(uintptr_t)workspace, 256));
  size_t off = nms_layout(B, pre, nb, &a.ws, base);
  a.boxes_all = reinterpret_cast<float4*>(base + off);
  off += align_up((size_t)B * count * 16, 256);
  a.score_all = reinterpret_cast<float*>(base + off);
  off += align_up((size_t)B * count * 4, 256);
  int P2 = 64;
  while (P2 < pre) P2 <<= 1;
  a.ghist = reinterpret_cast<int*>(base + off);  // B x 4 x 256 counters, then B candidate counters
  a.gncand = a.ghist + (size_t)B * 4 * 256;
  const size_t counters_bytes = (size_t)B * (4 * 256 + 1) * 4;
  off += align_up(counters_bytes, 256);
  a.gcand = reinterpret_cast<unsigned long long*>(base + off);
  off += align_up((size_t)B * P2 * 8, 256);
  const size_t need = off + (size_t)(base - (char*)workspace);
  if (!workspace || workspace_bytes < need)
    return fail(SD_ERR_WORKSPACE, "Proposal workspace too small: %zu < %zu bytes", workspace_bytes,
                need);
  proposal_v3_anchors(feature_stride, scales_host, n_scales, ratios_host, n_ratios, a.anchors);
  a.cls_prob = cls_prob; a.bbox_pred = bbox_pred; a.im_info = im_info;
  a.A = A; a.H = H; a.W = W; a.stride = feature_stride; a.count = count; a.pre = pre;
  a.min_size = (float)rpn_min_size;
  a.iou_loss = iou_loss ? 1 : 0;
  a.P2 = P2;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(proposal_decode_kernel, dim3((count + 255) / 256, B), dim3(256), 0, st, a);
  SD_LAUNCH_CHECK();
  const size_t lds = (size_t)P2 * sizeof(unsigned long long);
  // levels with many anchors: the select is spread over G workgroups per image (6 short launches);
  // small levels keep the single-workgroup kernel (knob proposal_topk: 1 forces it, 2 forces multi)
  const int mode = tuning("proposal_topk", 0);
  const bool multi = mode == 2 || (mode != 1 && count >= 32768);
  if (multi) {
    int G = (count + 2047) / 2048;  // one 8 x 256 trip per workgroup
    if (G > 128) G = 128;
    a.G = G;
    SD_HIP_CHECK(hipMemsetAsync(a.ghist, 0, counters_bytes, st));
    for (int pass = 0; pass < 4; ++pass)
      hipLaunchKernelGGL(topk_hist_kernel, dim3(G, B), dim3(256), 0, st, a, pass);
    hipLaunchKernelGGL(topk_compact_kernel, dim3(G, B), dim3(256), 0, st, a);
    if (lds > 64 * 1024)
      SD_HIP_CHECK(hipFuncSetAttribute((const void*)topk_finish_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(topk_finish_kernel, dim3(B), dim3(1024), lds, st, a);
  } else {
    if (lds > 64 * 1024)
      SD_HIP_CHECK(hipFuncSetAttribute((const void*)proposal_topk_kernel,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(proposal_topk_kernel, dim3(B), dim3(1024), lds, st, a);
  }
  SD_LAUNCH_CHECK();
  MaskArgs ma{a.ws, pre, nb, nb * (nb + 1) / 2, threshold, 1};  // IoU >= threshold (:319)
  hipLaunchKernelGGL(nms_mask_kernel, dim3((ma.npairs + 3) / 4, B), dim3(256), 0, st, ma);
  SD_LAUNCH_CHECK();
  ScanArgs ca{a.ws, out, score, nullptr, pre, post, nb, is_train ? 1 : 0};
  hipLaunchKernelGGL(nms_scan_kernel, dim3(B), dim3(64), 0, st, ca);
  SD_LAUNCH_CHECK();
  return SD_OK;
}

// get_top_proposal (models/FPN/get_top_proposal.py:15-39): bbox (B,N,4), score (B,N) -> the top_n
// rows by score (descending, ties keep the lower row), zero padded when N < top_n
struct TopArgs {
  const float4* bbox;
  const float* score;
  float4* out_bbox;
  float* out_score;
  int N, top_n, P2, select;
};

__global__ __launch_bounds__(1024) void top_proposal_kernel(TopArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned long long keys[];
  __shared__ int hist[260];
  __shared__ int ncand;
  const int img = blockIdx.x, tid = threadIdx.x, T = blockDim.x;
  const float* sc = a.score + (long)img * a.N;
  if (a.select) {  // top_n < N: select first, sort only the P2 >= top_n survivors
    select_sort_topk(sc, a.N, a.top_n, a.P2, keys, hist, &ncand);
  } else {
    for (int i = tid; i < a.P2; i += T)
      keys[i] = i < a.N ? (((unsigned long long)ordered_desc_bits(sc[i]) << 32) | (unsigned)i) : ~0ull;
    __syncthreads();
    bitonic_sort_lds(keys, a.P2, tid, T);
  }
  for (int i = tid; i < a.top_n; i += T) {
    float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
    float s = 0.f;
    if (i < a.N) {
      const int src = (int)(unsigned)(keys[i] & 0xffffffffu);
      b = a.bbox[(long)img * a.N + src];
      s = sc[src];
    }
    a.out_bbox[(long)img * a.top_n + i] = b;
    a.out_score[(long)img * a.top_n + i] = s;
  }
}

extern "C" int sd_proposal_v3(const float* cls_prob, const float* bbox_pred, const float* im_info,
                              float* out, float* score, int B, int A, int H, int W,
                              int rpn_pre_nms_top_n, int rpn_post_nms_top_n, float threshold,
                              int rpn_min_size, const float* scales_host, int n_scales,
                              const float* ratios_host, int n_ratios, int feature_stride,
                              int is_train, void* workspace, size_t workspace_bytes, void* stream) {
  return proposal_v3_impl(cls_prob, bbox_pred, im_info, out, score, B, A, H, W, rpn_pre_nms_top_n,
                          rpn_post_nms_top_n, threshold, rpn_min_size, scales_host, n_scales,
                          ratios_host, n_ratios, feature_stride, is_train, 0, workspace,
                          workspace_bytes, stream);
}

extern "C" int sd_proposal_v3_iou(const float* cls_prob, const float* bbox_pred,
                                  const float* im_info, float* out, float* score, int B, int A,
                                  int H, int W, int rpn_pre_nms_top_n, int rpn_post_nms_top_n,
                                  float threshold, int rpn_min_size, const float* scales_host,
                                  int n_scales, const float* ratios_host, int n_ratios,
                                  int feature_stride, int is_train, void* workspace,
                                  size_t workspace_bytes, void* stream) {
  return proposal_v3_impl(cls_prob, bbox_pred, im_info, out, score, B, A, H, W, rpn_pre_nms_top_n,
                          rpn_post_nms_top_n, threshold, rpn_min_size, scales_host, n_scales,
                          ratios_host, n_ratios, feature_stride, is_train, 1, workspace,
                          workspace_bytes, stream);
}

extern "C" int sd_get_top_proposal(const float* bbox, const float* score, int B, int N, int top_n,
                                   float* out_bbox, float* out_score, void* stream) {
  SD_REQUIRE(B >= 0 && N >= 0 && top_n >= 0, "negative dimension");
  if (B == 0 || top_n == 0) return SD_OK;
  SD_REQUIRE(out_bbox && out_score && ((bbox && score) || N == 0), "null tensor pointer");
  SD_REQUIRE((((uintptr_t)bbox | (uintptr_t)out_bbox) & 15) == 0, "bbox must be 16-byte aligned");
  int P2 = 64;
  while (P2 < N) P2 <<= 1;
  int P2t = 64;
  while (P2t < top_n) P2t <<= 1;
  // fewer rows wanted than given: radix select, then sort the survivors only (N is then bounded by
  // the 2^24 rows of the select, not by the LDS sort capacity)
  const int select = (top_n < N && P2t < P2) ? 1 : 0;
  if (select) P2 = P2t;
  SD_REQUIRE(P2 <= kMaxSortKeys, "get_top_proposal: N=%d / top_n=%d exceeds %d", N, top_n,
             kMaxSortKeys);
  SD_REQUIRE(N < (1 << 24), "get_top_proposal: N=%d >= 2^24", N);
  TopArgs a{reinterpret_cast<const float4*>(bbox), score, reinterpret_cast<float4*>(out_bbox),
            out_score, N, top_n, P2, select};
  const size_t lds = (size_t)P2 * sizeof(unsigned long long);
  if (lds > 64 * 1024)
    SD_HIP_CHECK(hipFuncSetAttribute((const void*)top_proposal_kernel,
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(top_proposal_kernel, dim3(B), dim3(1024), lds, (hipStream_t)stream, a);
  SD_LAUNCH_CHECK();
  return SD_OK;
}
