// ROIAlign_v2 for gfx950 (MI355X): what the three translation units share -- argument structs, the
// reference's per-element expressions, the band plan of the forward, the unit plan of the backward --
// and the host functions they call across files.
//   roi_align_fwd.hip    forward kernels (band-resident, tiled fallback, naive), their launcher, the forward
//                        entry points of the C ABI
//   roi_align_bwd.hip    backward kernels (fused wide-load kernel, four-plane C4 kernel, per-level planes,
//                        global atomics), their launchers, the backward entry points
//   roi_align_prep.hip   the rois-only pre-passes (forward item lists / tap entries, backward band lists /
//                        tap tables, and the merged launch of both)
//   roi_align_lists.h    the band-list builder the pre-pass and the workspace-free backward share
//
// Semantics follow the reference bit for bit (build with -ffp-contract=off, IEEE divide/sqrt):
//   forward   operator_cxx/contrib/roi_align_v2-inl.h:61-153 (max over the interior sample grid of
//             each bin, float argmax (x,y) stored); mixed float/double loop bounds kept (:120-125)
//   backward  operator_cxx/contrib/roi_align_v2.cu:35-84 (GPU scatter semantics)
//   assign    models/FPN/assign_layer_fpn.py:17-41
#pragma once
#include "common.h"
#include "../../include/simpledet_ops.h"
#include <float.h>
#include <math.h>
#include <type_traits>
#include <hip/hip_fp16.h>

namespace sd {

struct RoiLevels {
  const float* data[SD_MAX_FPN_LEVELS];
  int H[SD_MAX_FPN_LEVELS], W[SD_MAX_FPN_LEVELS], stride[SD_MAX_FPN_LEVELS];
  float scale[SD_MAX_FPN_LEVELS];
  int nlvl;
  float canon_scale, canon_level, k_min, k_max;
};

// models/FPN/assign_layer_fpn.py:27-33 in float32; returns level index or -1 (matches no stride)
__device__ __forceinline__ int fpn_level(float x1, float y1, float x2, float y2,
                                         const RoiLevels& L) {
  float area = (x2 - x1 + 1.f) * (y2 - y1 + 1.f);
  float s = sqrtf(area);
  float t = floorf(L.canon_level + log2f(s / L.canon_scale + 1e-6f));
  t = t < L.k_min ? L.k_min : (t > L.k_max ? L.k_max : t);
  if (!(t == t)) return -1;
  int ts = ((int)ldexpf(1.f, (int)t)) & 255;  // (2 ** lvl).astype('uint8')
  int lvl = -1;
  for (int l = L.nlvl - 1; l >= 0; --l)
    if (ts == L.stride[l]) lvl = l;
  return lvl;
}

// ------------------------------------------------------------------------------------------------
// exact per-element forward (reference structure)
// ------------------------------------------------------------------------------------------------
struct FwdOut {
  float val, ax, ay;
  int code;  // packed arg-max: row sample * 3 + column sample, 255 = none
};

template <typename TP = float>
__device__ __forceinline__ FwdOut roi_align_fwd_elem(const TP* __restrict__ plane, int height,
                                                     int width, float x1, float y1, float x2,
                                                     float y2, float spatial_scale, int ph, int pw,
                                                     int pooled_height, int pooled_width) {
  float roi_start_w = x1 * spatial_scale;
  float roi_start_h = y1 * spatial_scale;
  float roi_end_w = x2 * spatial_scale;
  float roi_end_h = y2 * spatial_scale;
  float roi_width = roi_end_w - roi_start_w;
  float roi_height = roi_end_h - roi_start_h;
  float bin_size_h = roi_height / (float)pooled_height;
  float bin_size_w = roi_width / (float)pooled_width;
  float hstart = (float)ph * bin_size_h;
  float wstart = (float)pw * bin_size_w;
  float hend = (float)(ph + 1) * bin_size_h;
  float wend = (float)(pw + 1) * bin_size_w;
  hstart = fminr(fmaxr(hstart + roi_start_h, 0.f), (float)(height - 1));
  hend = fminr(fmaxr(hend + roi_start_h, 0.f), (float)(height - 1));
  wstart = fminr(fmaxr(wstart + roi_start_w, 0.f), (float)(width - 1));
  wend = fminr(fmaxr(wend + roi_start_w, 0.f), (float)(width - 1));
  bool is_empty = (hend <= hstart) || (wend <= wstart);
  FwdOut o{0.f, -1.f, -1.f, 255};
  if (!is_empty) {
    o.val = -FLT_MAX;
    float h_stride = (float)((double)(hend - hstart) / 3.0);
    float w_stride = (float)((double)(wend - wstart) / 3.0);
    double hlim = (double)(hend - h_stride) + 0.01;
    double wlim = (double)(wend - w_stride) + 0.01;
    float hstep = fmaxr(h_stride, 0.01f), wstep = fmaxr(w_stride, 0.01f);
    int ik = 0;
    for (float h = hstart + h_stride; (double)h <= hlim; h += hstep, ++ik) {
      int hlow = iminr(imaxr((int)floorf(h), 0), height - 1);
      int hhigh = iminr(imaxr((int)ceilf(h), 0), height - 1);
      float alpha = (hlow == hhigh) ? 0.5f : (h - (float)hlow) / (float)(hhigh - hlow);
      int il = 0;
      for (float w = wstart + w_stride; (double)w <= wlim; w += wstep, ++il) {
        int wleft = iminr(imaxr((int)floorf(w), 0), width - 1);
        int wright = iminr(imaxr((int)ceilf(w), 0), width - 1);
        float beta = (wleft == wright) ? 0.5f : (w - (float)wleft) / (float)(wright - wleft);
        float value = (1 - alpha) * (1 - beta) * (float)plane[hlow * width + wleft] +
                      alpha * (1 - beta) * (float)plane[hhigh * width + wleft] +
                      (1 - alpha) * beta * (float)plane[hlow * width + wright] +
                      alpha * beta * (float)plane[hhigh * width + wright];
        if (value > o.val) {
          o.val = value;
          o.ax = w;
          o.ay = h;
          o.code = ik * 3 + il;
        }
      }
    }
  }
  return o;
}

// Packed arg-max rows: one byte per output, each (RoI, channel) row padded to whole dwords so that
// the backward fetches four codes with one aligned 4-byte load (7x7: 49 -> 52 bytes)
__host__ __device__ constexpr int amax_stride(int pp) { return (pp + 3) & ~3; }

// 8-byte load of two adjacent floats that is only 4-byte aligned
struct __attribute__((packed, aligned(4))) F2u {
  float x, y;
};

// 16-byte load of four adjacent floats that is only 4-byte aligned
struct __attribute__((packed, aligned(4))) F4u {
  float x, y, z, w;
};

struct FwdArgs {
  RoiLevels L;
  const float* rois;
  float* out;
  float* ax;
  float* ay;
  int B, C, R, PH, PW;
  unsigned char* amax8;  // packed arg-max output (fused op); when set, ax / ay are not written
  float* coords;         // with amax8: (B*R, 2, 3*P) sample-coordinate table the backward decodes with
  int nslice;  // channel slices per RoI (one workgroup each)
  int fbslice; // band kernel: channel slices of a RoI on its exact per-element path (one workgroup each)
  int ablate;  // profiling only: 1 stop after the tables
  long long* dbg;                     // profiling build only: per-wave phase clocks (or null)
  int half_io;                        // 1: the feature maps and `out` are fp16 (band kernel only)
};

// sample table of one axis bin; returns the number of loop iterations (reference loop, capped at 3)
// (float)((double)x / 3.0) == x / 3.0f exactly (double rounding is innocuous for one IEEE division
// when the wide format has >= 2p+2 bits), so the stride uses the float divide; high - low is 0 or 1
// so the reference's division by (high - low) is the identity.
__device__ __forceinline__ int axis_samples(int p, int pooled, float start_c, float end_c,
                                            float scale, int size, int mul, float* val, float* frac,
                                            int* off) {
  float roi_start = start_c * scale;
  float roi_end = end_c * scale;
  float roi_len = roi_end - roi_start;
  float bin = roi_len / (float)pooled;
  float lo = (float)p * bin;
  float hi = (float)(p + 1) * bin;
  lo = fminr(fmaxr(lo + roi_start, 0.f), (float)(size - 1));
  hi = fminr(fmaxr(hi + roi_start, 0.f), (float)(size - 1));
  int cnt = -1;
  off[0] = off[1] = off[2] = off[3] = -1;
  if (!(hi <= lo)) {
    cnt = 0;
    float stride = (hi - lo) / 3.0f;
    double lim = (double)(hi - stride) + 0.01;
    float step = fmaxr(stride, 0.01f);
    for (float v = lo + stride; (double)v <= lim; v += step) {
      if (cnt < 2) {
        int low = iminr(imaxr((int)floorf(v), 0), size - 1);
        int high = iminr(imaxr((int)ceilf(v), 0), size - 1);
        val[cnt] = v;
        frac[cnt] = (low == high) ? 0.5f : (v - (float)low);
        off[2 * cnt] = low * mul;
        off[2 * cnt + 1] = high * mul;
      }
      ++cnt;
      if (cnt >= 3) break;
    }
  }
  return cnt;
}

// words of fwd->bwd state per RoI and per pooled row/column: 3 sample coordinates + 3 (packed
// neighbours, fraction) pairs; layout per RoI: [3*(PH+PW) coordinates | 3*(PH+PW) pairs]
constexpr int kCoordWords = 9;

// coordinate of sample k of axis bin p: the same float expressions as axis_samples / the
// reference loop (start + stride, then += max(stride, 0.01f) per further sample), so a packed
// arg-max (k, l) decodes to exactly the float the forward would have stored
__device__ __forceinline__ float sample_coord(int p, int pooled, float start_c, float end_c,
                                              float scale, int size, int k) {
  const float roi_start = start_c * scale;
  const float roi_end = end_c * scale;
  const float roi_len = roi_end - roi_start;
  const float bin = roi_len / (float)pooled;
  float lo = (float)p * bin;
  float hi = (float)(p + 1) * bin;
  lo = fminr(fmaxr(lo + roi_start, 0.f), (float)(size - 1));
  hi = fminr(fmaxr(hi + roi_start, 0.f), (float)(size - 1));
  const float stride = (hi - lo) / 3.0f;
  const float step = fmaxr(stride, 0.01f);
  float v = lo + stride;
  for (int i = 0; i < k; ++i) v += step;
  return v;
}

// Backward-ready form of one table coordinate: the two clamped neighbour indices packed as
// lo | hi << 16 (-1: the coordinate is the "nothing pooled" sentinel) and the interpolation
// fraction, computed with exactly the expressions of the backward (floor / ceil / clamp,
// (v - lo) / (hi - lo), 0.5 when lo == hi).  The backward of the packed path then needs no
// floor, ceil, clamp or division per gradient element.
__device__ __forceinline__ void store_tap(float* dst, float v, int size) {
  int packed = -1;
  float frac = 0.f;
  if (v != -1.f) {
    const int lo = iminr(imaxr((int)floorf(v), 0), size - 1);
    const int hi = iminr(imaxr((int)ceilf(v), 0), size - 1);
    frac = (lo == hi) ? 0.5f : (v - (float)lo) / (float)(hi - lo);
    packed = lo | (hi << 16);
  }
  reinterpret_cast<int*>(dst)[0] = packed;
  dst[1] = frac;
}

// sample-coordinate table of the packed arg-max: coords[roi][0][p*3 + k] = row coordinate of sample
// k of bin row p, coords[roi][1][q*3 + l] = column coordinate (2 * 3 * P floats per RoI)
// ---- band-resident forward: constants and the plan its pre-pass and its kernel share ----
constexpr int kBandThreads = 1024, kBandWaves = kBandThreads / kWave;
constexpr int kBandBufFloats = 16896;   // 66 KB per buffer, two buffers per workgroup
constexpr int kBandHalo = 8;            // rows below a band that its items may still tap
constexpr int kBandMaxBands = 16;
constexpr int kBandNP = 4;              // passes a wave keeps in registers (one round)
constexpr int kBandMaxUnits = 512;      // virtual units ((level, image, band) x rounds of items) of one launch
constexpr int kBandFillCost = 140, kBandPlaneCost = 60, kBandSetupCost = 1500;  // cost model, in item times
constexpr int kBandSub = 4;             // list segments per unit (pre-pass workgroups per (level, image))
constexpr int kBandFallbackWGs = 0;    // (no separate exact-path blocks: the band workgroups do that work last)

typedef float v2f __attribute__((ext_vector_type(2)));

struct BandPlan {
  int owned[SD_MAX_FPN_LEVELS];      // rows a band owns (H: the whole plane is one band)
  int rows[SD_MAX_FPN_LEVELS];       // rows a band loads (owned + halo; H for whole planes)
  int nbands[SD_MAX_FPN_LEVELS];
  int g[SD_MAX_FPN_LEVELS];          // planes per fill (1 for banded levels)
  int unit_base[SD_MAX_FPN_LEVELS];  // first unit of the level; unit = base + img * nbands + band
  int halo[SD_MAX_FPN_LEVELS];       // rows below a band its items may still tap
  int nwg;                           // band workgroups (after the fallback workgroups)
  int nunits;
  int grab;                          // channels a workgroup reserves at a time
  int gbias;                         // per cent added to the cost estimate of multi-plane units
  int tail;                          // last per cent of a unit's channels handed out in small pieces
  int tail_planes;                   // planes of such a piece (<= G)
  int pool;
  uint4* rowent;    // [B*R][pool]  {lo0 | step0 << 20 | empty << 31, lo1 | step1 << 20, a0, a1}
  uint4* colent;    // [B*R][pool]  {left0 | dup0 << 12 | left1 << 13 | dup1 << 25 | empty << 26, -, b0, b1}
  float2* rowval;   // [B*R][pool]  sample coordinates (float arg-max outputs only)
  float2* colval;
  unsigned* items;  // [B][SD_MAX_FPN_LEVELS][kBandSub][ceil(R / kBandSub) * pool]  RoI | p << 16, by band
  int2* seg;        // [unit][kBandSub] {first item of the segment, items}
  unsigned char* fbflag;  // [B*R] 1: handled by the exact per-element workgroups, 2: constant output (nothing pooled)
  int* chan_ctr;    // [kBandMaxUnits] next channel of every virtual unit (zeroed by the pre-pass)
  int nlist, nent;  // pre-pass blocks: lists, entries (then, packed, the coordinate table)
};

struct BandArgs {
  FwdArgs f;
  BandPlan p;
};

// ---- backward ----
struct BwdArgs {
  RoiLevels L;        // data[] unused; used for the level filter
  const float* dy;
  const float* ax;
  const float* ay;
  const float* rois;
  float* dx;          // this level's gradient (B,C,H,W)
  int B, C, R, PP, H, W;
  float scale;
  int filter_lvl;     // >= 0: only RoIs assigned to this level contribute (fused FPN); -1: all
  int band_rows, nbands;
  int req;            // 1 write, 3 add
  int ablate;         // profiling only: 1 skip scatter, 2 skip write-out, 4 skip list build
};

// fused backward (every level of the pyramid in ONE launch): workgroup = (level, image, row band,
// channel); the band of the gradient plane lives in LDS and is written to HBM exactly once with 16-B
// stores: no zero-fill pass, no global atomics, no per-level launch boundary / tail.
struct BwdFusedArgs {
  RoiLevels L;
  const float* dy;
  const float* ax;
  const float* ay;
  const unsigned char* amax8;  // packed arg-max (k*3 + l, 255 none) instead of ax / ay
  const float* coords;         // with amax8: the forward's sample-coordinate table
  const float* rois;
  float* dx[SD_MAX_FPN_LEVELS];
  int band_rows[SD_MAX_FPN_LEVELS], nbands[SD_MAX_FPN_LEVELS];
  int block_end[SD_MAX_FPN_LEVELS];  // exclusive prefix of workgroups per level (in launch order)
  int order[SD_MAX_FPN_LEVELS];      // launch order of the levels (largest first)
  int nlaunch;
  int B, C, R, PP;
  int filter;  // 1: fused FPN (a RoI contributes to its assigned level only), 0: single level
  int req;
  int ablate;
  // packed4 with a workspace: the RoI lists of all (level, image, band) units, built once by
  // roi_align_bwd_lists instead of once per channel: unit u -> [count, weight bound, R indices]
  int* ws_list;
  float* ws_taps;                    // [unit][R][2 * 3 * (PH + PW)] band-relative tap entries, list order
  int unit_base[SD_MAX_FPN_LEVELS];  // first unit of launch-order level li
  int lists_units;                   // (level, image, band) units the list pre-pass covers
  int half_io;                       // dy and dx are fp16 (packed arg-max, wide kernel only)
  int float_adds;                    // 1: every workgroup sums with fp32 compare-and-swap adds (tuning key roi_align_bwd_fx = 0)
  int pix_bound_words;               // list pre-pass: LDS words for the per-cell (4 x 4 pixels) weight bound of a band, of the
                                     // largest level that gets one; 0: the bands keep the summed bound
};
constexpr int kPixBoundMaxWords = 10240;  // 40 KB of difference array at most

constexpr int kListSplit = 4;         // 512-thread blocks per unit of the stand-alone list pre-pass
constexpr int kMergedListSplit = 2;   // 1024-thread blocks per unit inside the merged pre-pass

// ---- host functions used across the translation units ----
int fill_levels(RoiLevels& L, const float* const* feats, const int* Hs, const int* Ws, const int* strides,
                int nlvl, float canon_scale, float canon_level);
int check_dims(int B, int C, int R, int ph, int pw);
// roi_align_fwd.hip
int launch_fwd(FwdArgs& a, hipStream_t st, void* workspace = nullptr, size_t workspace_bytes = 0,
               const BwdFusedArgs* bplan = nullptr, bool* bplan_done = nullptr);
// roi_align_bwd.hip
int launch_bwd_fused(BwdFusedArgs& a, int nlvl, hipStream_t st, void* workspace = nullptr,
                     size_t workspace_bytes = 0, int prepass = 0);
int launch_bwd(BwdArgs& a, hipStream_t st);
// roi_align_prep.hip: the forward's pre-pass (nblocks = list + entry + coordinate blocks), with the
// backward's list blocks behind it when bplan is given (one launch); the backward's own list pre-pass
int launch_fwd_prep(const BandArgs& A, int pool, int nblocks, const BwdFusedArgs* bplan, hipStream_t st);
int launch_bwd_lists(const BwdFusedArgs& a, int units, hipStream_t st);

}  // namespace sd
