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
//                         the diagonal word is resolved with scalar ops, one ballot per kept box.
//                         Kept boxes are written straight to out/score; the tail is zero padded.
// No host round trip, no device synchronisation, one stream.
#include "common.h"
#include "../../include/simpledet_ops.h"

namespace sd {

constexpr int kMaxSortKeys = 16384;

__device__ __forceinline__ unsigned ordered_desc_bits(float f) {
  unsigned u = __float_as_uint(f);
  u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);  // ascending-order-preserving map
  return ~u;                                        // descending
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
    for (int k = 2; k <= a.P2; k <<= 1) {
      for (int j = k >> 1; j > 0; j >>= 1) {
        for (int t = tid; t < a.P2 / 2; t += T) {
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
        __syncthreads();
      }
    }
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

// Transposed suppression matrix: T[col][rb] = bit r set  <=>  box (rb*64 + r) suppresses box col
// (IoU over the threshold, r ranked before col).  Lanes hold the 64 row boxes of block rb, the 64
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
  if (col < a.pre) a.ws.mask[((long)img * a.pre + col) * a.nb + rb] = word;
}

struct ScanArgs {
  NmsWs ws;
  float* out;
  float* score;
  int* keep_index;
  int pre, post, nb;
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
  for (int rb = 0; rb < a.nb; ++rb) {
    if (nkeep >= a.post) break;  // only the first `post` kept boxes are ever output
    const int c = rb * kWave + lane;
    const bool valid = c < a.pre;
    const unsigned long long* Tc = T + (long)(valid ? c : 0) * a.nb;
    unsigned long long acc = 0;
    for (int w0 = 0; w0 < rb; w0 += 16) {
      unsigned long long t[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) t[e] = (w0 + e < rb) ? Tc[w0 + e] : 0ull;
#pragma unroll
      for (int e = 0; e < 16; ++e)
        if (w0 + e < rb) acc |= t[e] & kw[w0 + e];
    }
    const unsigned long long diag = valid ? Tc[rb] : 0ull;
    unsigned long long cur = __ballot(acc != 0 || !valid);
    unsigned long long keepmask = 0;
    unsigned long long cand = ~cur;
    while (cand) {  // wave-uniform: one trip per kept box of this block
      const int k = __ffsll((long long)cand) - 1;
      keepmask |= 1ull << k;
      cur |= __ballot((diag >> k) & 1ull) | (1ull << k);
      cand = ~cur;
    }
    if (lane == 0) kw[rb] = keepmask;
    // PrepareOutput (nms.cu:207-233) for the boxes kept in this block
    if (keepmask & (1ull << lane)) {
      const int rank = nkeep + __popcll(keepmask & ((1ull << lane) - 1));
      if (rank < a.post) {
        reinterpret_cast<float4*>(out)[rank] = boxes[c];
        score[rank] = sscore[c];
        if (keep_index) keep_index[rank] = order[c];
      }
    }
    nkeep += __popcll(keepmask);
  }
  if (nkeep > a.post) nkeep = a.post;
  for (int i = nkeep + lane; i < a.post; i += kWave) {
    reinterpret_cast<float4*>(out)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    score[i] = 0.f;
    if (keep_index) keep_index[i] = -1;
  }
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
  if (!already_sorted && P2 > kMaxSortKeys)
    return fail(SD_ERR_UNSUPPORTED, "NMS: N=%d exceeds the in-LDS sort capacity (%d); pass "
                "already_sorted=1 with pre-sorted input", N, kMaxSortKeys);
  SD_REQUIRE(pre <= kMaxSortKeys, "NMS: pre_nms_top_n=%d exceeds %d", pre, kMaxSortKeys);
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
  hipLaunchKernelGGL(nms_sort_kernel, dim3(B), dim3(sort_threads), lds, st, sa);
  SD_LAUNCH_CHECK();

  MaskArgs ma{ws, pre, nb, nb * (nb + 1) / 2, threshold, threshold_ge};
  hipLaunchKernelGGL(nms_mask_kernel, dim3((ma.npairs + 3) / 4, B), dim3(256), 0, st, ma);
  SD_LAUNCH_CHECK();

  ScanArgs ca{ws, out, score, keep_index, pre, post, nb};
  hipLaunchKernelGGL(nms_scan_kernel, dim3(B), dim3(64), 0, st, ca);
  SD_LAUNCH_CHECK();
  return SD_OK;
}
