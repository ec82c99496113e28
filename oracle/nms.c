/*
 * ORACLE (test infrastructure, not product code) -- CPU restatement of the NMS family.
 *
 * Follows:
 *   _contrib_NMS  operator_cxx/contrib/nms.cu:92-202 (devIoU, nms_kernel, _nms host scan),
 *                 :207-233 (PrepareOutput), :249-365 (NMSGPUOp::Forward: stable sort by score
 *                 descending, pre_nms_top_n, post_nms_top_n).  The GPU path is the spec; the CPU
 *                 path nms.cc is broken.
 *   soft_nms      operator_py/cython/cpu_nms.pyx:98-203
 *   greedy_nms    operator_py/cython/cpu_nms.pyx:37-87
 *   bbox_overlaps operator_py/cython/bbox.pyx:31-72
 * Pinned against the reference's own Cython modules built into oracle/_ref (oracle/build_ref.py),
 * tests/test_nms.py.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline float fmaxc(float a, float b) { return a > b ? a : b; } /* CUDA max(float,float) */
static inline float fminc(float a, float b) { return a < b ? a : b; }

/* nms.cu:92-100 */
static float dev_iou(const float* a, const float* b) {
  float left = fmaxc(a[0], b[0]), right = fminc(a[2], b[2]);
  float top = fmaxc(a[1], b[1]), bottom = fminc(a[3], b[3]);
  float width = fmaxc(right - left + 1, 0.f), height = fmaxc(bottom - top + 1, 0.f);
  float interS = width * height;
  float Sa = (a[2] - a[0] + 1) * (a[3] - a[1] + 1);
  float Sb = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
  return interS / (Sa + Sb - interS);
}

/* stable merge sort of indices by score descending == thrust::stable_sort_by_key(greater<float>) */
static void stable_sort_desc(const float* score, int* order, int n) {
  int* tmp = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  for (int width = 1; width < n; width *= 2) {
    for (int lo = 0; lo < n; lo += 2 * width) {
      int mid = lo + width < n ? lo + width : n, hi = lo + 2 * width < n ? lo + 2 * width : n;
      int i = lo, j = mid, k = lo;
      while (i < mid && j < hi) {
        /* take from the right run only when strictly greater: keeps equal keys in input order */
        if (score[order[j]] > score[order[i]]) tmp[k++] = order[j++];
        else tmp[k++] = order[i++];
      }
      while (i < mid) tmp[k++] = order[i++];
      while (j < hi) tmp[k++] = order[j++];
    }
    memcpy(order, tmp, sizeof(int) * (size_t)n);
  }
  free(tmp);
}

void orc_nms(const float* dets, int B, int N, int pre_nms_top_n, int post_nms_top_n,
             float threshold, int already_sorted, float* out, float* score, int* keep_out) {
  /* nms.cu:274-277 */
  int pre = pre_nms_top_n > 0 ? pre_nms_top_n : N;
  if (pre > N) pre = N;
  int post = post_nms_top_n < pre ? post_nms_top_n : pre;
  const int col_blocks = (pre + 63) / 64;
  int* order = (int*)malloc(sizeof(int) * (size_t)(N + 1));
  float* sc = (float*)malloc(sizeof(float) * (size_t)(N + 1));
  float* boxes = (float*)malloc(sizeof(float) * 5 * (size_t)(pre + 1));
  uint64_t* mask = (uint64_t*)calloc((size_t)(pre + 1) * (col_blocks + 1), sizeof(uint64_t));
  uint64_t* remv = (uint64_t*)malloc(sizeof(uint64_t) * (size_t)(col_blocks + 1));
  int* keep = (int*)malloc(sizeof(int) * (size_t)(pre + 1));
  for (int b = 0; b < B; ++b) {
    const float* p = dets + (long)b * 5 * N;
    for (int i = 0; i < N; ++i) { /* CopyScoreKernel :63-73 */
      sc[i] = p[i * 5 + 4];
      order[i] = i;
    }
    if (!already_sorted) stable_sort_desc(sc, order, N); /* :304-312 */
    for (int i = 0; i < pre; ++i) memcpy(boxes + i * 5, p + (long)order[i] * 5, 5 * sizeof(float));
    /* nms_kernel :102-147 */
    for (int i = 0; i < pre; ++i)
      for (int cb = 0; cb < col_blocks; ++cb) {
        uint64_t t = 0;
        int col_size = pre - cb * 64 < 64 ? pre - cb * 64 : 64;
        int start = (i / 64 == cb) ? (i % 64) + 1 : 0;
        for (int j = start; j < col_size; ++j)
          if (dev_iou(boxes + i * 5, boxes + (cb * 64 + j) * 5) > threshold) t |= 1ULL << j;
        mask[(long)i * col_blocks + cb] = t;
      }
    /* _nms host scan :186-201 */
    memset(remv, 0, sizeof(uint64_t) * (size_t)col_blocks);
    int num_to_keep = 0;
    for (int i = 0; i < pre; ++i) {
      int nblock = i / 64, inblock = i % 64;
      if (!(remv[nblock] & (1ULL << inblock))) {
        keep[num_to_keep++] = i;
        for (int j = nblock; j < col_blocks; ++j) remv[j] |= mask[(long)i * col_blocks + j];
      }
    }
    /* PrepareOutput :207-233 (count = post, out_size = num_to_keep) */
    float* o = out + (long)b * 4 * post;
    float* s = score + (long)b * post;
    for (int i = 0; i < post; ++i) {
      if (i < num_to_keep) {
        memcpy(o + i * 4, boxes + keep[i] * 5, 4 * sizeof(float));
        s[i] = boxes[keep[i] * 5 + 4];
        if (keep_out) keep_out[(long)b * post + i] = order[keep[i]];
      } else {
        o[i * 4 + 0] = o[i * 4 + 1] = o[i * 4 + 2] = o[i * 4 + 3] = 0.0f;
        s[i] = 0;
        if (keep_out) keep_out[(long)b * post + i] = -1;
      }
    }
  }
  free(order); free(sc); free(boxes); free(mask); free(remv); free(keep);
}

/* cpu_nms.pyx:27-31 */
static inline float pmax(float a, float b) { return a >= b ? a : b; }
static inline float pmin(float a, float b) { return a <= b ? a : b; }

/* cpu_nms.pyx:98-203.  boxes (n,5) is the working copy (boxes_in.copy(), :106). */
int orc_soft_nms(float* boxes, int64_t* inds, int n, float sigma, float Nt, float threshold,
                 unsigned method) {
  unsigned N = (unsigned)n;
  float iw, ih, ua, maxscore, tx1, tx2, ty1, ty2, ts, area, weight, ov, x1, x2, y1, y2;
  int pos, maxpos;
  for (int i = 0; i < n; ++i) inds[i] = i; /* :113 */
  const int N0 = n;                         /* range(N) is evaluated once (:115) */
  for (int i = 0; i < N0; ++i) {
    if ((unsigned)i >= N) break; /* i >= N: boxes[i] reads stale rows and every while is a no-op
                                    on the kept prefix -- see below */
    maxscore = boxes[i * 5 + 4];
    maxpos = i;
    tx1 = boxes[i * 5 + 0]; ty1 = boxes[i * 5 + 1]; tx2 = boxes[i * 5 + 2]; ty2 = boxes[i * 5 + 3];
    ts = boxes[i * 5 + 4];
    int64_t ti = inds[i];
    pos = i + 1;
    while ((unsigned)pos < N) { /* :129-133 */
      if (maxscore < boxes[pos * 5 + 4]) {
        maxscore = boxes[pos * 5 + 4];
        maxpos = pos;
      }
      pos = pos + 1;
    }
    /* :136-150 swap */
    for (int k = 0; k < 5; ++k) boxes[i * 5 + k] = boxes[maxpos * 5 + k];
    inds[i] = inds[maxpos];
    boxes[maxpos * 5 + 0] = tx1; boxes[maxpos * 5 + 1] = ty1; boxes[maxpos * 5 + 2] = tx2;
    boxes[maxpos * 5 + 3] = ty2; boxes[maxpos * 5 + 4] = ts;
    inds[maxpos] = ti;
    tx1 = boxes[i * 5 + 0]; ty1 = boxes[i * 5 + 1]; tx2 = boxes[i * 5 + 2]; ty2 = boxes[i * 5 + 3];
    ts = boxes[i * 5 + 4];
    (void)ts;
    pos = i + 1;
    while ((unsigned)pos < N) { /* :161-201 */
      x1 = boxes[pos * 5 + 0]; y1 = boxes[pos * 5 + 1]; x2 = boxes[pos * 5 + 2]; y2 = boxes[pos * 5 + 3];
      /* Cython turns the Python int literal 1 next to a C float into the C double literal 1.0
       * (see the generated C: "(x2 - x1) + 1.0"), so every "+ 1" is a double add and the products
       * of two such terms are double products, narrowed to float only on assignment */
      area = (float)(((double)(x2 - x1) + 1.0) * ((double)(y2 - y1) + 1.0));
      iw = (float)((double)(pmin(tx2, x2) - pmax(tx1, x1)) + 1.0);
      if (iw > 0) {
        ih = (float)((double)(pmin(ty2, y2) - pmax(ty1, y1)) + 1.0);
        if (ih > 0) {
          ua = (float)(((((double)(tx2 - tx1) + 1.0) * ((double)(ty2 - ty1) + 1.0)) + (double)area) -
                       (double)(iw * ih)); /* float(...) :171 */
          ov = iw * ih / ua;
          if (method == 1) weight = ov > Nt ? (float)(1.0 - (double)ov) : 1;
          else if (method == 2)
            weight = (float)exp((double)(-(ov * ov) / sigma)); /* np.exp on a Python float :180 */
          else weight = ov > Nt ? 0 : 1;
          /* :187 Python-float weight times np.float32 score, stored to the fp32 array: the exact
             double product of two floats rounded once == the fp32 product */
          boxes[pos * 5 + 4] = weight * boxes[pos * 5 + 4];
          if (boxes[pos * 5 + 4] < threshold) { /* :191-199 */
            for (int k = 0; k < 5; ++k) boxes[pos * 5 + k] = boxes[(N - 1) * 5 + k];
            inds[pos] = inds[N - 1];
            N = N - 1;
            pos = pos - 1;
          }
        }
      }
      pos = pos + 1;
    }
  }
  return (int)N;
}

/* cpu_nms.pyx:37-87; order = scores.argsort()[::-1] is numpy's (unstable) introsort reversed:
 * ties between equal scores are implementation-defined there; here ties keep the later index
 * first (what a stable ascending sort reversed gives).  Tests use distinct scores. */
int orc_greedy_nms(const float* dets, int n, float thresh, int64_t* keep) {
  int* order = (int*)malloc(sizeof(int) * (size_t)(n + 1));
  float* neg = (float*)malloc(sizeof(float) * (size_t)(n + 1));
  char* suppressed = (char*)calloc((size_t)n + 1, 1);
  float* areas = (float*)malloc(sizeof(float) * (size_t)(n + 1));
  for (int i = 0; i < n; ++i) {
    order[i] = n - 1 - i;
    neg[i] = dets[i * 5 + 4];
    areas[i] = (dets[i * 5 + 2] - dets[i * 5 + 0] + 1) * (dets[i * 5 + 3] - dets[i * 5 + 1] + 1);
  }
  /* stable ascending sort, then reversed == stable descending over reversed input order */
  stable_sort_desc(neg, order, n);
  for (int _i = 0; _i < n; ++_i) {
    int i = order[_i];
    if (suppressed[i]) continue;
    float ix1 = dets[i * 5], iy1 = dets[i * 5 + 1], ix2 = dets[i * 5 + 2], iy2 = dets[i * 5 + 3];
    float iarea = areas[i];
    for (int _j = _i + 1; _j < n; ++_j) {
      int j = order[_j];
      if (suppressed[j]) continue;
      float xx1 = pmax(ix1, dets[j * 5]), yy1 = pmax(iy1, dets[j * 5 + 1]);
      float xx2 = pmin(ix2, dets[j * 5 + 2]), yy2 = pmin(iy2, dets[j * 5 + 3]);
      float w = pmax(0.0f, (float)((double)(xx2 - xx1) + 1.0)); /* "+ 1" is a double add */
      float h = pmax(0.0f, (float)((double)(yy2 - yy1) + 1.0));
      float inter = w * h;
      float ovr = inter / (iarea + areas[j] - inter);
      if (ovr >= thresh) suppressed[j] = 1;
    }
  }
  int cnt = 0;
  for (int i = 0; i < n; ++i)
    if (!suppressed[i]) keep[cnt++] = i; /* np.where(suppressed == 0)[0]: ascending index */
  free(order); free(neg); free(suppressed); free(areas);
  return cnt;
}

/* Cython's builtin min/max on C floats: min(a, b) -> (b < a) ? b : a ; max(a, b) -> (b > a) ? b : a */
static inline float bmin(float a, float b) { return b < a ? b : a; }
static inline float bmax(float a, float b) { return b > a ? b : a; }

/* bbox.pyx:31-72 */
void orc_bbox_overlaps(const float* boxes, int n, const float* query, int k, float* overlaps) {
  memset(overlaps, 0, sizeof(float) * (size_t)n * k);
  for (int kk = 0; kk < k; ++kk) {
    const float* q = query + kk * 4;
    /* "+ 1" is a double add in the generated C, as in soft_nms above */
    float box_area = (float)(((double)(q[2] - q[0]) + 1.0) * ((double)(q[3] - q[1]) + 1.0));
    for (int nn = 0; nn < n; ++nn) {
      const float* b = boxes + nn * 4;
      float iw = (float)((double)(bmin(b[2], q[2]) - bmax(b[0], q[0])) + 1.0);
      if (iw > 0) {
        float ih = (float)((double)(bmin(b[3], q[3]) - bmax(b[1], q[1])) + 1.0);
        if (ih > 0) {
          float ua = (float)(((((double)(b[2] - b[0]) + 1.0) * ((double)(b[3] - b[1]) + 1.0)) +
                              (double)box_area) - (double)(iw * ih));
          overlaps[nn * k + kk] = iw * ih / ua;
        }
      }
    }
  }
}
