/*
 * ORACLE (test infrastructure, not product code) -- CPU restatement of _contrib_DecodeBBox and of
 * the test-time per-class detection filter that feeds soft-NMS.
 *
 * Follows:
 *   decode  operator_cxx/contrib/decodebbox.cc:34-80 (BBoxTransformXYWH), :84-131
 *           (BBoxTransformXYXY); shapes decodebbox-inl.h:85-107; class_agnostic decodes with the
 *           deltas of class 1 (decodebbox.cc:56)
 *   filter  detection_test.py:233-247 (do_nms: per class, score > min_det_score, [box, score] rows)
 *
 * exp(): decodebbox.cc calls the unqualified exp() on a float in a host .cc file, which resolves
 * to the C double exp(double); the product with the float width is therefore a DOUBLE multiply
 * narrowed on assignment.  Oracle and kernel both use (float)(exp((double)dw) * (double)width).
 */
#include "oracle.h"
#include <math.h>
#include <string.h>

static inline float fmax2(float a, float b) { return a < b ? b : a; } /* std::max */
static inline float fmin2(float a, float b) { return b < a ? b : a; } /* std::min */

void orc_decode_bbox(const float* rois, const float* deltas, const float* im_info, float* out,
                     int B, int R, int K, const float* means, const float* stds,
                     int class_agnostic, int xyxy) {
  const int num_class = class_agnostic ? 1 : K;
  for (int n = 0; n < B; ++n)
    for (int index = 0; index < R; ++index)
      for (int cls = 0; cls < num_class; ++cls) {
        const float* b = rois + ((long)n * R + index) * 4;
        const int decode_cls = class_agnostic ? 1 : cls;
        const float* d = deltas + ((long)n * R + index) * 4 * K + decode_cls * 4;
        float width = b[2] - b[0] + 1.0f;
        float height = b[3] - b[1] + 1.0f;
        float px1, py1, px2, py2;
        if (!xyxy) {
          float ctr_x = b[0] + 0.5f * (width - 1.0f);
          float ctr_y = b[1] + 0.5f * (height - 1.0f);
          float dx = d[0] * stds[0] + means[0];
          float dy = d[1] * stds[1] + means[1];
          float dw = d[2] * stds[2] + means[2];
          float dh = d[3] * stds[3] + means[3];
          float pred_ctr_x = dx * width + ctr_x;
          float pred_ctr_y = dy * height + ctr_y;
          float pred_w = (float)(exp((double)dw) * (double)width);
          float pred_h = (float)(exp((double)dh) * (double)height);
          px1 = pred_ctr_x - 0.5f * (pred_w - 1.0f);
          py1 = pred_ctr_y - 0.5f * (pred_h - 1.0f);
          px2 = pred_ctr_x + 0.5f * (pred_w - 1.0f);
          py2 = pred_ctr_y + 0.5f * (pred_h - 1.0f);
        } else {
          float dx1 = d[0] * stds[0] + means[0];
          float dy1 = d[1] * stds[1] + means[1];
          float dx2 = d[2] * stds[2] + means[2];
          float dy2 = d[3] * stds[3] + means[3];
          px1 = b[0] + dx1 * width;
          py1 = b[1] + dy1 * height;
          px2 = b[2] + dx2 * width;
          py2 = b[3] + dy2 * height;
        }
        float* o = out + ((long)n * R + index) * 4 * num_class + cls * 4;
        o[0] = fmax2(fmin2(px1, im_info[n * 3 + 1] - 1.0f), 0.0f);
        o[1] = fmax2(fmin2(py1, im_info[n * 3 + 0] - 1.0f), 0.0f);
        o[2] = fmax2(fmin2(px2, im_info[n * 3 + 1] - 1.0f), 0.0f);
        o[3] = fmax2(fmin2(py2, im_info[n * 3 + 0] - 1.0f), 0.0f);
      }
}

/* detection_test.py:236-247: problem (n, cid) = rows with score > min_det_score, in row order.
 * bbox (B,R,4*Kb) with Kb == K (class specific) or 1 (shared); dets (B*K, R, 5); counts (B*K). */
void orc_det_filter(const float* bbox, const float* cls_score, int B, int R, int K, int Kb,
                    float min_det_score, float* dets, int* counts) {
  for (int n = 0; n < B; ++n)
    for (int cid = 0; cid < K; ++cid) {
      float* d = dets + ((long)(n * K + cid) * R) * 5;
      int m = 0;
      for (int r = 0; r < R; ++r) {
        const float s = cls_score[((long)n * R + r) * K + cid];
        if (s > min_det_score) {
          const float* b = bbox + ((long)n * R + r) * 4 * Kb + (Kb == 1 ? 0 : cid * 4);
          memcpy(d + m * 5, b, 4 * sizeof(float));
          d[m * 5 + 4] = s;
          ++m;
        }
      }
      counts[n * K + cid] = m;
    }
}
