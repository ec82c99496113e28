/*
 * ORACLE (test infrastructure, not product code) -- CPU restatement of _contrib_Proposal_v3 (GPU
 * path = the spec) and of the FPN get_top_proposal CustomOp.
 *
 * Follows:
 *   anchors     operator_cxx/contrib/proposal_v3-inl.h:279-318 (float math, floor(size/ratio),
 *               ratio-major)
 *   grid        operator_cxx/contrib/proposal_v3.cu:64-85   (ProposalGridKernel)
 *   decode      :92-155 (BBoxPredKernel: ctr = x1 + 0.5 w, x2 = ... - 1, dw/dh clipped at
 *               log(1000/16), clip to the image)
 *   top-k       :136-153 of Forward (thrust::stable_sort_by_key descending, first pre_nms_top_n)
 *   filter      :211-235 (FilterBoxKernel, AFTER the top-k: small boxes are enlarged and get score -1
 *               but keep their place in the list)
 *   nms         :271-381 (IoU >= threshold, +1 area convention, greedy in list order)
 *   output      :386-416 (PrepareOutput: zero padding at test time, cyclic repeat when is_train)
 *   get_top     models/FPN/get_top_proposal.py:15-39 (argsort descending, first top_n)
 *
 * One deliberate deviation: the reference evaluates exp(dw) with CUDA's expf, whose last-bit
 * behaviour no other libm reproduces; both this oracle and the HIP kernel use
 * (float)exp((double)dw), which is the correctly rounded value in all but ~1e-8 of cases.
 * iou_loss = true (IoUPredKernel :163-205: corners = anchor corners + deltas, clipped) is
 * orc_proposal_v3_iou; no config of the reference enables it.
 */
#include "oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

static inline float fmaxc(float a, float b) { return a > b ? a : b; } /* CUDA max / min */
static inline float fminc(float a, float b) { return a < b ? a : b; }

/* proposal_v3-inl.h:292-318 */
void orc_proposal_v3_anchors(int feature_stride, const float* scales, int ns, const float* ratios,
                             int nr, float* anchors /* nr*ns*4 */) {
  float base[4] = {0.0f, 0.0f, (float)(feature_stride - 1.0), (float)(feature_stride - 1.0)};
  int n = 0;
  for (int j = 0; j < nr; ++j)
    for (int k = 0; k < ns; ++k) {
      const float scale = scales[k], ratio = ratios[j];
      float w = base[2] - base[0] + 1.0f;
      float h = base[3] - base[1] + 1.0f;
      float x_ctr = (float)(base[0] + 0.5 * (w - 1.0f));
      float y_ctr = (float)(base[1] + 0.5 * (h - 1.0f));
      float size = w * h;
      float size_ratios = floorf(size / ratio);
      float new_w = rintf(sqrtf(size_ratios)) * scale;
      float new_h = rintf((new_w / scale * ratio)) * scale;
      anchors[n * 4 + 0] = x_ctr - 0.5f * (new_w - 1.0f);
      anchors[n * 4 + 1] = y_ctr - 0.5f * (new_h - 1.0f);
      anchors[n * 4 + 2] = x_ctr + 0.5f * (new_w - 1.0f);
      anchors[n * 4 + 3] = y_ctr + 0.5f * (new_h - 1.0f);
      ++n;
    }
}

static float iou_plus1(const float* a, const float* b) {
  float left = fmaxc(a[0], b[0]), right = fminc(a[2], b[2]);
  float top = fmaxc(a[1], b[1]), bottom = fminc(a[3], b[3]);
  float width = fmaxc(right - left + 1, 0.f), height = fmaxc(bottom - top + 1, 0.f);
  float interS = width * height;
  float Sa = (a[2] - a[0] + 1) * (a[3] - a[1] + 1);
  float Sb = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
  return interS / (Sa + Sb - interS);
}

static void stable_sort_desc(const float* score, int* order, int n) {
  int* tmp = (int*)malloc(sizeof(int) * (size_t)(n > 0 ? n : 1));
  for (int width = 1; width < n; width *= 2) {
    for (int lo = 0; lo < n; lo += 2 * width) {
      int mid = lo + width < n ? lo + width : n, hi = lo + 2 * width < n ? lo + 2 * width : n;
      int i = lo, j = mid, k = lo;
      while (i < mid && j < hi) {
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

int orc_proposal_v3_post(int count, int pre_nms_top_n, int post_nms_top_n, int is_train) {
  int pre = pre_nms_top_n > 0 ? pre_nms_top_n : count;
  if (pre > count) pre = count;
  int post = post_nms_top_n < pre ? post_nms_top_n : pre;
  if (!is_train) post = post_nms_top_n;
  return post;
}

static void proposal_v3_impl(const float* cls_prob, const float* bbox_pred, const float* im_info, int B,
                             int A, int H, int W, int pre_nms_top_n, int post_nms_top_n, float threshold,
                             int min_size, const float* scales, int ns, const float* ratios, int nr,
                             int feature_stride, int is_train, int iou_loss, float* out,
                             float* score_out) {
  const int count = A * H * W;
  int pre = pre_nms_top_n > 0 ? pre_nms_top_n : count;
  if (pre > count) pre = count;
  const int post = orc_proposal_v3_post(count, pre_nms_top_n, post_nms_top_n, is_train);
  float anchors[4 * 256];
  orc_proposal_v3_anchors(feature_stride, scales, ns, ratios, nr, anchors);
  float* prop = (float*)malloc(sizeof(float) * 5 * (size_t)(count + 1));
  float* sc = (float*)malloc(sizeof(float) * (size_t)(count + 1));
  int* order = (int*)malloc(sizeof(int) * (size_t)(count + 1));
  float* top = (float*)malloc(sizeof(float) * 5 * (size_t)(pre + 1));
  char* removed = (char*)malloc((size_t)pre + 1);
  int* keep = (int*)malloc(sizeof(int) * (size_t)(pre + 1));
  for (int b = 0; b < B; ++b) {
    const float* fg = cls_prob + ((long)b * 2 * A + A) * H * W; /* foreground half */
    const float* deltas = bbox_pred + (long)b * 4 * A * H * W;
    const float im_height = im_info[b * 3 + 0], im_width = im_info[b * 3 + 1];
    const float im_scale = im_info[b * 3 + 2];
    for (int index = 0; index < count; ++index) {
      const int a = index % A, w = (index / A) % W, h = index / A / W;
      /* ProposalGridKernel */
      float x1 = anchors[a * 4 + 0] + w * feature_stride;
      float y1 = anchors[a * 4 + 1] + h * feature_stride;
      float x2 = anchors[a * 4 + 2] + w * feature_stride;
      float y2 = anchors[a * 4 + 3] + h * feature_stride;
      /* BBoxPredKernel */
      float width = x2 - x1 + 1.0f;
      float height = y2 - y1 + 1.0f;
      float ctr_x = x1 + 0.5f * width;
      float ctr_y = y1 + 0.5f * height;
      float dx = deltas[((long)(a * 4) * H + h) * W + w];
      float dy = deltas[((long)(a * 4 + 1) * H + h) * W + w];
      float dw = deltas[((long)(a * 4 + 2) * H + h) * W + w];
      float dh = deltas[((long)(a * 4 + 3) * H + h) * W + w];
      dw = (float)((double)dw < 4.135166556742356 ? (double)dw : 4.135166556742356);
      dh = (float)((double)dh < 4.135166556742356 ? (double)dh : 4.135166556742356);
      float pred_ctr_x = dx * width + ctr_x;
      float pred_ctr_y = dy * height + ctr_y;
      float pred_w = (float)exp((double)dw) * width;
      float pred_h = (float)exp((double)dh) * height;
      float px1 = pred_ctr_x - 0.5f * pred_w;
      float py1 = pred_ctr_y - 0.5f * pred_h;
      float px2 = pred_ctr_x + 0.5f * pred_w - 1.0f;
      float py2 = pred_ctr_y + 0.5f * pred_h - 1.0f;
      if (iou_loss) { /* IoUPredKernel, proposal_v3.cu:181-194: the four deltas move the corners */
        px1 = x1 + deltas[((long)(a * 4) * H + h) * W + w];
        py1 = y1 + deltas[((long)(a * 4 + 1) * H + h) * W + w];
        px2 = x2 + deltas[((long)(a * 4 + 2) * H + h) * W + w];
        py2 = y2 + deltas[((long)(a * 4 + 3) * H + h) * W + w];
      }
      prop[index * 5 + 0] = fmaxc(fminc(px1, im_width - 1.0f), 0.0f);
      prop[index * 5 + 1] = fmaxc(fminc(py1, im_height - 1.0f), 0.0f);
      prop[index * 5 + 2] = fmaxc(fminc(px2, im_width - 1.0f), 0.0f);
      prop[index * 5 + 3] = fmaxc(fminc(py2, im_height - 1.0f), 0.0f);
      prop[index * 5 + 4] = fg[((long)a * H + h) * W + w];
      /* IoUPredKernel only (:201-203; commented out in BBoxPredKernel :151-153): anchors past the
       * unpadded image, real_height = (int)(im_height / feature_stride) (:510-511) */
      if (iou_loss && (h >= (int)(im_height / feature_stride) || w >= (int)(im_width / feature_stride)))
        prop[index * 5 + 4] = -1.0f;
      sc[index] = prop[index * 5 + 4];
      order[index] = index;
    }
    stable_sort_desc(sc, order, count);
    for (int i = 0; i < pre; ++i) memcpy(top + i * 5, prop + (long)order[i] * 5, 5 * sizeof(float));
    /* FilterBoxKernel */
    for (int i = 0; i < pre; ++i) {
      float* d = top + i * 5;
      float ws_orig_scale = (d[2] - d[0]) / im_scale + 1.0f;
      float hs_orig_scale = (d[3] - d[1]) / im_scale + 1.0f;
      float min_size_max = fmaxc((float)min_size, 1.0f);
      float ws = d[2] - d[0] + 1.0f;
      float hs = d[3] - d[1] + 1.0f;
      float x_ctr = d[0] + ws / 2.0f;
      float y_ctr = d[1] + hs / 2.0f;
      if (ws_orig_scale < min_size_max || hs_orig_scale < min_size_max || x_ctr >= im_width ||
          y_ctr >= im_height) {
        d[0] -= min_size_max / 2;
        d[1] -= min_size_max / 2;
        d[2] += min_size_max / 2;
        d[3] += min_size_max / 2;
        d[4] = -1.0f;
      }
    }
    /* greedy NMS in list order, IoU >= threshold */
    memset(removed, 0, (size_t)pre);
    int nkeep = 0;
    for (int i = 0; i < pre; ++i) {
      if (removed[i]) continue;
      keep[nkeep++] = i;
      for (int j = i + 1; j < pre; ++j)
        if (!removed[j] && iou_plus1(top + i * 5, top + j * 5) >= threshold) removed[j] = 1;
    }
    /* PrepareOutput */
    float* o = out + (long)b * 4 * post;
    float* s = score_out + (long)b * post;
    for (int index = 0; index < post; ++index) {
      if (index < nkeep) {
        memcpy(o + index * 4, top + keep[index] * 5, 4 * sizeof(float));
        s[index] = top[keep[index] * 5 + 4];
      } else if (is_train && nkeep > 0) {
        const int k = keep[index % nkeep];
        memcpy(o + index * 4, top + k * 5, 4 * sizeof(float));
        s[index] = top[k * 5 + 4];
      } else {
        o[index * 4 + 0] = o[index * 4 + 1] = o[index * 4 + 2] = o[index * 4 + 3] = 0.0f;
        s[index] = 0;
      }
    }
  }
  free(prop); free(sc); free(order); free(top); free(removed); free(keep);
}

void orc_proposal_v3(const float* cls_prob, const float* bbox_pred, const float* im_info, int B,
                     int A, int H, int W, int pre_nms_top_n, int post_nms_top_n, float threshold,
                     int min_size, const float* scales, int ns, const float* ratios, int nr,
                     int feature_stride, int is_train, float* out, float* score_out) {
  proposal_v3_impl(cls_prob, bbox_pred, im_info, B, A, H, W, pre_nms_top_n, post_nms_top_n, threshold,
                   min_size, scales, ns, ratios, nr, feature_stride, is_train, 0, out, score_out);
}

void orc_proposal_v3_iou(const float* cls_prob, const float* bbox_pred, const float* im_info, int B,
                         int A, int H, int W, int pre_nms_top_n, int post_nms_top_n, float threshold,
                         int min_size, const float* scales, int ns, const float* ratios, int nr,
                         int feature_stride, int is_train, float* out, float* score_out) {
  proposal_v3_impl(cls_prob, bbox_pred, im_info, B, A, H, W, pre_nms_top_n, post_nms_top_n, threshold,
                   min_size, scales, ns, ratios, nr, feature_stride, is_train, 1, out, score_out);
}

/* models/FPN/get_top_proposal.py:15-39; ties keep the lower row first (stable) */
void orc_get_top_proposal(const float* bbox, const float* score, int B, int N, int top_n,
                          float* out_bbox, float* out_score) {
  int* order = (int*)malloc(sizeof(int) * (size_t)(N + 1));
  for (int b = 0; b < B; ++b) {
    for (int i = 0; i < N; ++i) order[i] = i;
    stable_sort_desc(score + (long)b * N, order, N);
    for (int i = 0; i < top_n; ++i) {
      if (i < N) {
        memcpy(out_bbox + ((long)b * top_n + i) * 4, bbox + ((long)b * N + order[i]) * 4, 16);
        out_score[(long)b * top_n + i] = score[(long)b * N + order[i]];
      } else {
        memset(out_bbox + ((long)b * top_n + i) * 4, 0, 16);
        out_score[(long)b * top_n + i] = 0;
      }
    }
  }
  free(order);
}
