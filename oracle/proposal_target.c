/*
 * ORACLE (test infrastructure, not product code) -- CPU restatement of ProposalTarget.
 *
 * Follows (DType = float, index_t = unsigned):
 *   op        operator_cxx/proposal_target-inl.h:123-256 (ProposalTargetOp::Forward: gt clean-up
 *             :155-162, roi clean-up + gt append :169-187, fg_rois_per_image truncation :194)
 *   sampling  operator_cxx/proposal_target.cc:21-163 (SampleROI), :165-185 (BBoxOverlap),
 *             :187-202 (ExpandBboxRegressionTargets), :204-227 (NonLinearTransformAndNormalization)
 *   RNG       std::random_shuffle (libstdc++ bits/stl_algo.h: for i in [1,n): swap(a[i],
 *             a[rand() % (i+1)])) over libc rand() (glibc random_r.c TYPE_3: degree 31, separation 3,
 *             r[i] = r[i-3] + r[i-31], output >> 1, 310 warm-up draws after srandom_r's LCG fill)
 * The RNG restatement is pinned against the real libc rand() and the real std::random_shuffle
 * (oracle/shuffle_ref.cc) in tests/test_proposal_target.py.
 *
 * Reference undefined behaviour, defined here: an image with no valid gt box makes
 * the reference read IOUs[i][0] out of bounds; here such an image yields max_overlap 0 / label 0
 * for every roi and the function returns -1.  Output rows past kept_indexes.size() stay zero.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---- glibc rand(): stdlib/random_r.c (TYPE_3) ---- */
void orc_glibc_srand(orc_glibc_rand* st, unsigned seed) {
  int32_t* r = st->r;
  if (seed == 0) seed = 1; /* __srandom_r: "We must make sure the seed is not 0" */
  r[0] = (int32_t)seed;
  int32_t word = (int32_t)seed;
  for (int i = 1; i < 31; ++i) { /* word = 16807 * hi ... Schrage, no overflow */
    long hi = word / 127773;
    long lo = word % 127773;
    long w = 16807 * lo - 2836 * hi;
    if (w < 0) w += 2147483647;
    word = (int32_t)w;
    r[i] = word;
  }
  st->f = 3; /* fptr = &state[rand_sep] */
  st->b = 0; /* rptr = &state[0] */
  for (int i = 0; i < 310; ++i) (void)orc_glibc_rand_next(st); /* kc = rand_deg * 10 discards */
}

int orc_glibc_rand_next(orc_glibc_rand* st) {
  uint32_t* r = (uint32_t*)st->r;
  uint32_t val = r[st->f] += r[st->b];
  int result = (int)(val >> 1);
  if (++st->f >= 31) st->f = 0;
  if (++st->b >= 31) st->b = 0;
  return result;
}

typedef int (*rand_fn)(void*);
static int rand_state(void* p) { return orc_glibc_rand_next((orc_glibc_rand*)p); }
static int rand_libc(void* p) { (void)p; return rand(); }

/* libstdc++ std::random_shuffle(first, last) */
static void shuffle(unsigned* a, size_t n, rand_fn rf, void* rs) {
  for (size_t i = 1; i < n; ++i) {
    size_t j = (size_t)(rf(rs) % (int)(i + 1));
    if (i != j) { unsigned t = a[i]; a[i] = a[j]; a[j] = t; }
  }
}

static inline float fmin2(float a, float b) { return b < a ? b : a; } /* std::min */
static inline float fmax2(float a, float b) { return a < b ? b : a; } /* std::max */

static int sample_roi(const float* all_rois, unsigned n_rois, const float* gt_boxes, unsigned n_gt,
                      const orc_proposal_target_param* p, unsigned fg_rois_per_image,
                      unsigned rois_per_image, rand_fn rf, void* rs, float* rois, float* labels,
                      float* bbox_targets, float* bbox_weights, float* match_gt_ious,
                      int* kept_out, unsigned* fg_this_out, unsigned* gt_of_row_out) {
  int ub = 0;
  float* max_overlaps = (float*)calloc(n_rois + 1, sizeof(float));
  float* all_labels = (float*)calloc(n_rois + 1, sizeof(float));
  unsigned* gt_assignment = (unsigned*)calloc(n_rois + 1, sizeof(unsigned));
  if (n_gt == 0) {
    ub = -1;
  } else {
    /* BBoxOverlap :165-185 + row arg-max :47-61 (first maximum wins: strict <) */
    float* iou = (float*)calloc((size_t)n_rois * n_gt + 1, sizeof(float));
    for (unsigned j = 0; j < n_gt; ++j) {
      const float* q = gt_boxes + j * 5;
      float query_box_area = (q[2] - q[0] + 1.f) * (q[3] - q[1] + 1.f);
      for (unsigned i = 0; i < n_rois; ++i) {
        const float* b = all_rois + i * 4;
        float iw = fmin2(b[2], q[2]) - fmax2(b[0], q[0]) + 1.f;
        if (iw > 0) {
          float ih = fmin2(b[3], q[3]) - fmax2(b[1], q[1]) + 1.f;
          if (ih > 0) {
            float box_area = (b[2] - b[0] + 1.f) * (b[3] - b[1] + 1.f);
            float union_area = box_area + query_box_area - iw * ih;
            iou[(size_t)i * n_gt + j] = iw * ih / union_area;
          }
        }
      }
    }
    for (unsigned i = 0; i < n_rois; ++i) {
      float max_value = iou[(size_t)i * n_gt];
      unsigned max_index = 0;
      for (unsigned j = 1; j < n_gt; ++j)
        if (max_value < iou[(size_t)i * n_gt + j]) {
          max_value = iou[(size_t)i * n_gt + j];
          max_index = j;
        }
      gt_assignment[i] = max_index;
      max_overlaps[i] = max_value;
      all_labels[i] = gt_boxes[max_index * 5 + 4];
    }
    free(iou);
  }
  /* :67-79 */
  unsigned* fg = (unsigned*)malloc(sizeof(unsigned) * (n_rois + 1));
  unsigned* neg = (unsigned*)malloc(sizeof(unsigned) * (n_rois + 1));
  unsigned* bg = (unsigned*)malloc(sizeof(unsigned) * (n_rois + 1));
  unsigned* kept = (unsigned*)malloc(sizeof(unsigned) * (rois_per_image + n_rois + 1));
  size_t nfg = 0, nneg = 0, nbg = 0, nkept = 0;
  for (unsigned i = 0; i < n_rois; ++i) {
    if (max_overlaps[i] >= p->fg_thresh) fg[nfg++] = i;
    else neg[nneg++] = i;
  }
  /* :81-86 */
  unsigned fg_this = fg_rois_per_image < nfg ? fg_rois_per_image : (unsigned)nfg;
  if (nfg > fg_this) {
    shuffle(fg, nfg, rf, rs);
    nfg = fg_this;
  }
  /* :94-104 */
  for (unsigned i = 0; i < n_rois; ++i)
    if (max_overlaps[i] >= p->bg_thresh_lo && max_overlaps[i] < p->bg_thresh_hi) bg[nbg++] = i;
  unsigned want_bg = rois_per_image - fg_this;
  unsigned bg_this = want_bg < nbg ? want_bg : (unsigned)nbg;
  if (nbg > bg_this) {
    shuffle(bg, nbg, rf, rs);
    nbg = bg_this;
  }
  for (unsigned i = 0; i < fg_this; ++i) kept[nkept++] = fg[i];
  for (unsigned i = 0; i < bg_this; ++i) kept[nkept++] = bg[i];
  /* :116-122 */
  while (nkept < rois_per_image && nneg > 0) {
    unsigned gap = rois_per_image - (unsigned)nkept;
    shuffle(neg, nneg, rf, rs);
    for (unsigned idx = 0; idx < gap && idx < nneg; ++idx) kept[nkept++] = neg[idx];
  }
  /* :128-135 */
  for (size_t i = 0; i < nkept; ++i) {
    if (i < fg_this) labels[i] = all_labels[kept[i]];
    memcpy(rois + i * 4, all_rois + (size_t)kept[i] * 4, 4 * sizeof(float));
    match_gt_ious[i] = max_overlaps[kept[i]];
    if (kept_out) kept_out[i] = (int)kept[i];
    if (gt_of_row_out) gt_of_row_out[i] = gt_assignment[kept[i]];
  }
  if (fg_this_out) *fg_this_out = fg_this;
  /* :138-161: targets for every output row; rows >= nkept read garbage in the reference and end up
   * zero because their label is 0, so they are skipped here */
  const unsigned K4 = 4 * (unsigned)p->num_classes;
  for (size_t i = 0; i < nkept && n_gt > 0; ++i) {
    const float* ex = rois + i * 4;
    const float* gt = gt_boxes + (size_t)gt_assignment[kept[i]] * 5;
    /* NonLinearTransformAndNormalization :204-227; "0.5 *" is a double expression */
    float ex_width = ex[2] - ex[0] + 1.f;
    float ex_height = ex[3] - ex[1] + 1.f;
    float ex_ctr_x = (float)(ex[0] + 0.5 * (ex_width - 1.f));
    float ex_ctr_y = (float)(ex[1] + 0.5 * (ex_height - 1.f));
    float gt_width = gt[2] - gt[0] + 1.f;
    float gt_height = gt[3] - gt[1] + 1.f;
    float gt_ctr_x = (float)(gt[0] + 0.5 * (gt_width - 1.f));
    float gt_ctr_y = (float)(gt[1] + 0.5 * (gt_height - 1.f));
    float t[4];
    t[0] = (gt_ctr_x - ex_ctr_x) / (ex_width + 1e-14f);
    t[1] = (gt_ctr_y - ex_ctr_y) / (ex_height + 1e-14f);
    t[2] = logf(gt_width / ex_width); /* using std::log; log(float) -> float overload */
    t[3] = logf(gt_height / ex_height);
    for (int k = 0; k < 4; ++k) t[k] -= p->bbox_mean[k];
    for (int k = 0; k < 4; ++k) t[k] /= p->bbox_std[k];
    float cls_f = labels[i];
    if (p->class_agnostic) cls_f = labels[i] < 1.f ? labels[i] : 1.f; /* :151-154 */
    if (cls_f > 0) { /* ExpandBboxRegressionTargets :187-202 */
      unsigned cls = (unsigned)cls_f;
      unsigned start = 4 * cls;
      if (start + 4 <= K4) {
        memcpy(bbox_targets + i * K4 + start, t, 4 * sizeof(float));
        memcpy(bbox_weights + i * K4 + start, p->bbox_weight, 4 * sizeof(float));
      }
    }
  }
  free(max_overlaps); free(all_labels); free(gt_assignment);
  free(fg); free(neg); free(bg); free(kept);
  return ub;
}

/* valid_ranges != NULL selects ProposalTarget_v2 (operator_cxx/proposal_target_v2-inl.h:128-266):
 *   - with filter_scales a gt box is appended to the candidate rois only if its area w*h lies in
 *     [valid_min^2, valid_max^2] (:188-203); the IoU / label side still sees every valid gt box
 *   - an image without candidate rois gets one all-zero roi, an image without valid gt one
 *     all-zero gt row (:244-249).  (The reference builds that row 4 wide and then reads column 4
 *     as the class: undefined; defined here as class 0.)
 *   image_rois == -1 ("keep every roi") makes the reference allocate (B, -1, .) host tensors
 *   (:209-213) and is rejected.  SampleROI itself (proposal_target_v2.cc:21-177) equals v1's for
 *   image_rois != -1. */
/* convertPoly2Mask, operator_cxx/proposal_mask_target.cc:148-216.  poly = [category, n_seg,
 * len_1..len_n, x,y,x,y,...]; the polygon is mapped into the RoI's mask_size x mask_size frame (in
 * DType = float arithmetic), x and y are swapped on the way into rleFrPoly so that its
 * column-major decode comes out row-major, and the segments are OR-ed. */
#include "mxshim/coco_api/common/maskApi.h"
static void convert_poly2mask(const float* roi, const float* poly, int mask_size, float* mask) {
  float w = roi[2] - roi[0], h = roi[3] - roi[1];
  w = fmax2(1.f, w);
  h = fmax2(1.f, h);
  int n_seg = (int)poly[1];
  if (n_seg < 0) n_seg = 0; /* the -1 filled row of an image without gt (:...-inl.h) */
  int offset = 2 + n_seg;
  RLE* rles;
  rlesInit(&rles, (siz)n_seg);
  for (int i = 0; i < n_seg; i++) {
    int cur_len = (int)poly[i + 2];
    double* xys = (double*)malloc(sizeof(double) * (cur_len > 0 ? cur_len : 1));
    for (int j = 0; j < cur_len; j++) {
      if (j % 2 == 0) xys[j] = (poly[offset + j + 1] - roi[1]) * mask_size / h;
      else xys[j] = (poly[offset + j - 1] - roi[0]) * mask_size / w;
    }
    rleFrPoly(rles + i, xys, (siz)(cur_len / 2), (siz)mask_size, (siz)mask_size);
    free(xys);
    offset += cur_len;
  }
  const int area = mask_size * mask_size;
  byte* bm = (byte*)malloc((size_t)area * (n_seg > 0 ? n_seg : 1));
  rleDecode(rles, bm, (siz)n_seg);
  for (int j = 0; j < area; j++) {
    float cur = 0;
    for (int i = 0; i < n_seg; i++)
      if (bm[(size_t)i * area + j] == 1) { cur = 1; break; }
    mask[j] = cur;
  }
  rlesFree(&rles, (siz)n_seg);
  free(bm);
}

/* convertPoly2MaskWithRatio, operator_cxx/proposal_mask_target.cc:20-152 (output_ratio = true, the
 * mask-scoring R-CNN heads: models/msrcnn/builder.py:219-237).  Besides the mask it rasterises the
 * polygon twice at image resolution -- inside the RoI (crop_h x crop_w, RoI corners truncated to
 * int) and inside the bounding box of RoI and polygon (full_h x full_w) -- and returns
 * max(crop_pixels / (full_pixels + 1e-4), 1e-10).  NOTE the mask itself is computed here with
 * DOUBLE coordinates (`double poly_index`, :53-63) where convertPoly2Mask uses DType. */
static double convert_poly2mask_with_ratio(const float* roi, const float* poly, int mask_size,
                                           float* mask) {
  float w = roi[2] - roi[0], h = roi[3] - roi[1];
  w = fmax2(1.f, w);
  h = fmax2(1.f, h);
  int n_seg = (int)poly[1];
  if (n_seg < 0) n_seg = 0;
  RLE *rles, *rles_origin, *rles_crop;
  rlesInit(&rles, (siz)n_seg);
  rlesInit(&rles_origin, (siz)n_seg);
  rlesInit(&rles_crop, (siz)n_seg);
  const int roi_x_1 = (int)roi[0], roi_x_2 = (int)roi[2], roi_y_1 = (int)roi[1], roi_y_2 = (int)roi[3];
  const int crop_w = roi_x_2 - roi_x_1 + 1, crop_h = roi_y_2 - roi_y_1 + 1;
  int offset = 2 + n_seg;
  double origin_x_1 = roi[0], origin_x_2 = roi[2], origin_y_1 = roi[1], origin_y_2 = roi[3];
  for (int i = 0; i < n_seg; i++) {
    int cur_len = (int)poly[i + 2];
    double* xys = (double*)malloc(sizeof(double) * (cur_len > 0 ? cur_len : 1));
    double* xys_crop = (double*)malloc(sizeof(double) * (cur_len > 0 ? cur_len + 1 : 2));
    for (int j = 0; j < cur_len; j++) { /* (an odd cur_len writes one past the reference's buffer) */
      if (j % 2 == 0) {
        double poly_index = poly[offset + j + 1];
        origin_y_1 = origin_y_1 < poly_index ? origin_y_1 : poly_index;
        origin_y_2 = origin_y_2 > poly_index ? origin_y_2 : poly_index;
        xys[j] = (poly_index - roi[1]) * mask_size / h;
        xys_crop[j + 1] = poly_index - roi[1];
      } else {
        double poly_index = poly[offset + j - 1];
        origin_x_1 = origin_x_1 < poly_index ? origin_x_1 : poly_index;
        origin_x_2 = origin_x_2 > poly_index ? origin_x_2 : poly_index;
        xys[j] = (poly_index - roi[0]) * mask_size / w;
        xys_crop[j - 1] = poly_index - roi[0];
      }
    }
    rleFrPoly(rles + i, xys, (siz)(cur_len / 2), (siz)mask_size, (siz)mask_size);
    rleFrPoly(rles_crop + i, xys_crop, (siz)(cur_len / 2), (siz)crop_h, (siz)crop_w);
    free(xys);
    free(xys_crop);
    offset += cur_len;
  }
  const int int_x_1 = (int)origin_x_1, int_x_2 = (int)origin_x_2, int_y_1 = (int)origin_y_1,
            int_y_2 = (int)origin_y_2;
  const int full_w = int_x_2 - int_x_1 + 1, full_h = int_y_2 - int_y_1 + 1;
  offset = 2 + n_seg;
  for (int i = 0; i < n_seg; i++) {
    int cur_len = (int)poly[i + 2];
    double* xys_origin = (double*)malloc(sizeof(double) * (cur_len > 0 ? cur_len + 1 : 2));
    for (int j = 0; j < cur_len; j++) {
      if (j % 2 == 0) xys_origin[j + 1] = (double)poly[offset + j + 1] - origin_y_1;
      else xys_origin[j - 1] = (double)poly[offset + j - 1] - origin_x_1;
    }
    rleFrPoly(rles_origin + i, xys_origin, (siz)(cur_len / 2), (siz)full_h, (siz)full_w);
    free(xys_origin);
    offset += cur_len;
  }
  const int area = mask_size * mask_size;
  const size_t nfull = (size_t)full_w * full_h, ncrop = (size_t)crop_w * crop_h;
  byte* bm = (byte*)malloc((size_t)area * (n_seg > 0 ? n_seg : 1));
  byte* bo = (byte*)calloc(nfull * (n_seg > 0 ? n_seg : 1) + 1, 1);
  byte* bc = (byte*)calloc(ncrop * (n_seg > 0 ? n_seg : 1) + 1, 1);
  rleDecode(rles, bm, (siz)n_seg);
  rleDecode(rles_origin, bo, (siz)n_seg);
  rleDecode(rles_crop, bc, (siz)n_seg);
  for (int j = 0; j < area; j++) {
    float cur = 0;
    for (int i = 0; i < n_seg; i++)
      if (bm[(size_t)i * area + j] == 1) { cur = 1; break; }
    mask[j] = cur;
  }
  double origin_mask_sum = 0, crop_mask_sum = 0;
  for (size_t i = 0; i < nfull; i++)
    for (int sgi = 0; sgi < n_seg; sgi++)
      if (bo[(size_t)sgi * nfull + i] == 1) { origin_mask_sum += 1; break; }
  for (size_t i = 0; i < ncrop; i++)
    for (int sgi = 0; sgi < n_seg; sgi++)
      if (bc[(size_t)sgi * ncrop + i] == 1) { crop_mask_sum += 1; break; }
  double mask_ratio = crop_mask_sum / (origin_mask_sum + 0.0001);
  mask_ratio = mask_ratio > 1e-10 ? mask_ratio : 1e-10;
  rlesFree(&rles, (siz)n_seg);
  rlesFree(&rles_origin, (siz)n_seg);
  rlesFree(&rles_crop, (siz)n_seg);
  free(bm); free(bo); free(bc);
  return mask_ratio;
}

/* valid_ranges selects the v2 / mask-target front end; gt_polys != NULL adds ProposalMaskTarget's
 * mask output (proposal_mask_target-inl.h:141-330, proposal_mask_target.cc:218-378):
 * mask_target (B, FG, ms, ms) with FG = (index_t)(image_rois * fg_fraction), -1 filled, rows
 * [0, fg_rois_this_image) rasterised from the polygon of the RoI's assigned gt box. */
static int proposal_target_impl(const float* rois, const float* gt_boxes, const float* valid_ranges,
                                int filter_scales, int N, int M,
                                const orc_proposal_target_param* p, rand_fn rf, void* rs,
                                float* roi_out, float* label, float* bbox_target,
                                float* bbox_weight, float* match_gt_iou, int* kept_index,
                                const float* gt_polys, int L, int mask_size, float* mask_target,
                                float* mask_ratio /* non-NULL: output_ratio = true */) {
  const int B = p->batch_images, S = p->image_rois, K4 = 4 * p->num_classes;
  int rc = 0;
  /* outputs are zero-initialised containers (proposal_target-inl.h:189-193) */
  memset(roi_out, 0, sizeof(float) * (size_t)B * S * 4);
  memset(label, 0, sizeof(float) * (size_t)B * S);
  memset(bbox_target, 0, sizeof(float) * (size_t)B * S * K4);
  memset(bbox_weight, 0, sizeof(float) * (size_t)B * S * K4);
  memset(match_gt_iou, 0, sizeof(float) * (size_t)B * S);
  if (kept_index)
    for (long i = 0; i < (long)B * S; ++i) kept_index[i] = -1;
  unsigned fg_rois_per_image = (unsigned)(S * p->fg_fraction); /* :194 static_cast truncation */
  float* kept_rois = (float*)malloc(sizeof(float) * 4 * (size_t)(N + M + 1));
  float* kept_gt = (float*)malloc(sizeof(float) * 5 * (size_t)(M + 1));
  const int v2 = valid_ranges != NULL || gt_polys != NULL;
  const int FG = (int)(S * p->fg_fraction);
  int* gt_src = (int*)malloc(sizeof(int) * (size_t)(M + 1));       /* kept gt -> input row */
  unsigned* gt_of_row = (unsigned*)malloc(sizeof(unsigned) * (size_t)(S + 1));
  if (mask_target)
    for (size_t t = 0; t < (size_t)B * FG * mask_size * mask_size; ++t) mask_target[t] = -1.f;
  for (int i = 0; i < B; ++i) {
    unsigned n_gt = 0, n_rois = 0;
    for (int j = 0; j < M; ++j) { /* :155-162 */
      const float* g = gt_boxes + ((size_t)i * M + j) * 5;
      if (g[4] != -1) { gt_src[n_gt] = j; memcpy(kept_gt + 5 * n_gt++, g, 5 * sizeof(float)); }
    }
    for (int j = 0; j < N; ++j) { /* :171-175, y2 == 0 indicates padding */
      const float* r = rois + ((size_t)i * N + j) * 4;
      if (r[3] > 0) memcpy(kept_rois + 4 * n_rois++, r, 4 * sizeof(float));
    }
    if (!p->proposal_without_gt) { /* :177-185 */
      float vmin = 0.f, vmax = 0.f;
      if (valid_ranges) {
        vmin = valid_ranges[2 * i] * valid_ranges[2 * i];
        vmax = valid_ranges[2 * i + 1] * valid_ranges[2 * i + 1];
      }
      for (unsigned j = 0; j < n_gt; ++j) {
        const float* g = kept_gt + 5 * j;
        if (valid_ranges && filter_scales) { /* DType w = x2 - x1 + 1.0 (double add, narrowed) */
          float w = (float)((double)(g[2] - g[0]) + 1.0), h = (float)((double)(g[3] - g[1]) + 1.0);
          if (w * h < vmin || w * h > vmax) continue;
        }
        memcpy(kept_rois + 4 * n_rois++, g, 4 * sizeof(float));
      }
    }
    int no_gt = 0;
    if (v2) { /* v2 :244-249, mask target -inl.h:268-276 */
      if (n_rois == 0) { memset(kept_rois, 0, 4 * sizeof(float)); n_rois = 1; }
      if (n_gt == 0) { memset(kept_gt, 0, 5 * sizeof(float)); n_gt = 1; no_gt = 1; }
    }
    unsigned fg_this = 0;
    int e = sample_roi(kept_rois, n_rois, kept_gt, n_gt, p, fg_rois_per_image, (unsigned)S, rf, rs,
                       roi_out + (size_t)i * S * 4, label + (size_t)i * S,
                       bbox_target + (size_t)i * S * K4, bbox_weight + (size_t)i * S * K4,
                       match_gt_iou + (size_t)i * S, kept_index ? kept_index + (size_t)i * S : NULL,
                       &fg_this, gt_of_row);
    if (e) rc = e;
    if (mask_target && !no_gt) /* an image without gt has one all -1 polygon row: nothing to draw
                                  (and no roi reaches fg_thresh against the 1-pixel zero box ... if one
                                  does, n_seg = -1 draws an all-zero mask; restated in the else branch) */
      for (unsigned r = 0; r < fg_this && r < (unsigned)FG; ++r) {
        const float* roi = roi_out + ((size_t)i * S + r) * 4;
        const float* poly = gt_polys + ((size_t)i * M + gt_src[gt_of_row[r]]) * L;
        float* m = mask_target + ((size_t)i * FG + r) * mask_size * mask_size;
        if (mask_ratio) /* proposal_mask_target.cc:368-372 */
          mask_ratio[(size_t)i * FG + r] = (float)convert_poly2mask_with_ratio(roi, poly, mask_size, m);
        else
          convert_poly2mask(roi, poly, mask_size, m);
      }
    else if (mask_target) {
      float neg1[4] = {-1.f, -1.f, -1.f, -1.f};
      for (unsigned r = 0; r < fg_this && r < (unsigned)FG; ++r) {
        const float* roi = roi_out + ((size_t)i * S + r) * 4;
        float* m = mask_target + ((size_t)i * FG + r) * mask_size * mask_size;
        if (mask_ratio)
          mask_ratio[(size_t)i * FG + r] = (float)convert_poly2mask_with_ratio(roi, neg1, mask_size, m);
        else
          convert_poly2mask(roi, neg1, mask_size, m);
      }
    }
  }
  free(kept_rois); free(kept_gt); free(gt_src); free(gt_of_row);
  return rc;
}

int orc_proposal_target(const float* rois, const float* gt_boxes, int N, int M,
                        const orc_proposal_target_param* p, orc_glibc_rand* rng, float* roi_out,
                        float* label, float* bbox_target, float* bbox_weight, float* match_gt_iou,
                        int* kept_index) {
  return proposal_target_impl(rois, gt_boxes, NULL, 0, N, M, p, rand_state, rng, roi_out, label,
                              bbox_target, bbox_weight, match_gt_iou, kept_index, NULL, 0, 0, NULL, NULL);
}

/* ProposalMaskTarget without output_ratio.  valid_ranges may be NULL (num_args = 3). */
int orc_proposal_mask_target(const float* rois, const float* gt_boxes, const float* gt_polys,
                             const float* valid_ranges, int filter_scales, int N, int M, int L,
                             int mask_size, const orc_proposal_target_param* p, orc_glibc_rand* rng,
                             float* roi_out, float* label, float* bbox_target, float* bbox_weight,
                             float* match_gt_iou, int* kept_index, float* mask_target) {
  if (p->image_rois < 0 || !gt_polys || !mask_target) return -2;
  return proposal_target_impl(rois, gt_boxes, valid_ranges, valid_ranges ? filter_scales : 0, N, M, p,
                              rand_state, rng, roi_out, label, bbox_target, bbox_weight,
                              match_gt_iou, kept_index, gt_polys, L, mask_size, mask_target, NULL);
}

/* ProposalMaskTarget with output_ratio = true: the seventh output mask_ratio (B, FG), zero filled
 * (proposal_mask_target-inl.h:244), rows [0, fg_rois_this_image) computed. */
int orc_proposal_mask_target_ratio(const float* rois, const float* gt_boxes, const float* gt_polys,
                                   const float* valid_ranges, int filter_scales, int N, int M, int L,
                                   int mask_size, const orc_proposal_target_param* p,
                                   orc_glibc_rand* rng, float* roi_out, float* label,
                                   float* bbox_target, float* bbox_weight, float* match_gt_iou,
                                   int* kept_index, float* mask_target, float* mask_ratio) {
  if (p->image_rois < 0 || !gt_polys || !mask_target || !mask_ratio) return -2;
  const int FG = (int)(p->image_rois * p->fg_fraction);
  memset(mask_ratio, 0, sizeof(float) * (size_t)p->batch_images * (FG > 0 ? FG : 0));
  return proposal_target_impl(rois, gt_boxes, valid_ranges, valid_ranges ? filter_scales : 0, N, M, p,
                              rand_state, rng, roi_out, label, bbox_target, bbox_weight,
                              match_gt_iou, kept_index, gt_polys, L, mask_size, mask_target,
                              mask_ratio);
}

int orc_poly2mask_ratio(const float* roi, const float* poly, int mask_size, float* mask, double* ratio) {
  *ratio = convert_poly2mask_with_ratio(roi, poly, mask_size, mask);
  return 0;
}

int orc_poly2mask(const float* roi, const float* poly, int mask_size, float* mask) {
  convert_poly2mask(roi, poly, mask_size, mask);
  return 0;
}

int orc_proposal_target_v2(const float* rois, const float* gt_boxes, const float* valid_ranges,
                           int filter_scales, int N, int M, const orc_proposal_target_param* p,
                           orc_glibc_rand* rng, float* roi_out, float* label, float* bbox_target,
                           float* bbox_weight, float* match_gt_iou, int* kept_index) {
  if (p->image_rois < 0 || !valid_ranges) return -2;
  return proposal_target_impl(rois, gt_boxes, valid_ranges, filter_scales, N, M, p, rand_state, rng,
                              roi_out, label, bbox_target, bbox_weight, match_gt_iou, kept_index,
                              NULL, 0, 0, NULL, NULL);
}

int orc_proposal_target_libc(const float* rois, const float* gt_boxes, int N, int M,
                             const orc_proposal_target_param* p, float* roi_out, float* label,
                             float* bbox_target, float* bbox_weight, float* match_gt_iou,
                             int* kept_index) {
  return proposal_target_impl(rois, gt_boxes, NULL, 0, N, M, p, rand_libc, NULL, roi_out, label,
                              bbox_target, bbox_weight, match_gt_iou, kept_index, NULL, 0, 0, NULL, NULL);
}
