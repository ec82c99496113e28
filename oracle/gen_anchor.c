/*
 * ORACLE (test infrastructure, not product code) -- CPU restatement of _contrib_GenAnchor.
 *
 * Follows:
 *   base anchors  operator_cxx/contrib/generate_anchor-inl.h:140-181 (_MakeAnchor, _Transform,
 *                 GenerateAnchors; DType = double, ratio-major enumeration :176-180)
 *   grid          operator_cxx/contrib/generate_anchor.cc:62-82 == generate_anchor.cu:61-81
 *                 (index = (h*W + w)*A + a, double add, cast to fp32 last)
 * Pinned by the reference's numpy twin (symbol/builder.py:904-938) through tests/golden/.
 */
#include "oracle.h"
#include <math.h>

void orc_gen_base_anchors(int feature_stride, const double* scales, int ns, const double* ratios,
                          int nr, double* base) {
  /* generate_anchor.cc:62-64: {0, 0, stride - 1.0f, stride - 1.0f} (float expression -> double) */
  const double b0 = 0.0f, b1 = 0.0f;
  const double b2 = (double)((float)feature_stride - 1.0f), b3 = (double)((float)feature_stride - 1.0f);
  int n = 0;
  for (int j = 0; j < nr; ++j)
    for (int k = 0; k < ns; ++k) {
      const double scale = scales[k], ratio = ratios[j];
      /* _Transform, generate_anchor-inl.h:153-168 (1.0f / 0.5f / 0.5 literals promote to double) */
      double w = b2 - b0 + 1.0f;
      double h = b3 - b1 + 1.0f;
      double x_ctr = b0 + 0.5 * (w - 1.0f);
      double y_ctr = b1 + 0.5 * (h - 1.0f);
      double size = w * h;
      double size_ratios = size / ratio;
      double new_w = rint(sqrt(size_ratios)) * scale;
      double new_h = rint((new_w / scale * ratio)) * scale;
      /* _MakeAnchor :142-151 */
      base[n * 4 + 0] = x_ctr - 0.5f * (new_w - 1.0f);
      base[n * 4 + 1] = y_ctr - 0.5f * (new_h - 1.0f);
      base[n * 4 + 2] = x_ctr + 0.5f * (new_w - 1.0f);
      base[n * 4 + 3] = y_ctr + 0.5f * (new_h - 1.0f);
      ++n;
    }
}

void orc_gen_anchor(float* out, int H, int W, int feature_stride, const double* scales, int ns,
                    const double* ratios, int nr) {
  double base[4 * 256];
  const int A = ns * nr;
  orc_gen_base_anchors(feature_stride, scales, ns, ratios, nr, base);
  for (int i = 0; i < A; ++i)
    for (int j = 0; j < H; ++j)
      for (int k = 0; k < W; ++k) {
        long index = (long)j * (W * A) + (long)k * A + i; /* generate_anchor.cc:74 */
        out[index * 4 + 0] = (float)(base[i * 4 + 0] + k * feature_stride);
        out[index * 4 + 1] = (float)(base[i * 4 + 1] + j * feature_stride);
        out[index * 4 + 2] = (float)(base[i * 4 + 2] + k * feature_stride);
        out[index * 4 + 3] = (float)(base[i * 4 + 3] + j * feature_stride);
      }
}
