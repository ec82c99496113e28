// _contrib_DecodeBBox and the test-time per-class detection filter, gfx950.
//   reference: operator_cxx/contrib/decodebbox.cc:34-131 (CPU only: the op copies rois/deltas/
//              im_info to the host, decodes in a triple loop, copies back), shapes
//              decodebbox-inl.h:85-107; detection_test.py:233-247 (numpy filter per class inside a
//              process pool, followed by the Cython soft-NMS).
// Both are pure streaming ops.  decode: one lane per (roi, class) box, 16-byte loads/stores.
// filter: one wave per (image, class) problem compacts the rows with score > min_det_score in row
// order (ballot + prefix popcount) straight into the (P, Nmax, 5) layout sd_soft_nms_batched reads,
// so bbox head output -> decode -> filter -> soft-NMS never leaves the device.
#include "common.h"
#include "../../include/simpledet_ops.h"
#include <math.h>

namespace sd {

struct DecodeArgs {
  const float* rois;
  const float* deltas;
  const float* im_info;
  float* out;
  int B, R, K, num_class, class_agnostic, xyxy;
  float means[4], stds[4];
};

__global__ __launch_bounds__(256) void decode_bbox_kernel(DecodeArgs a) {
  const long count = (long)a.B * a.R * a.num_class;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (long)gridDim.x * blockDim.x) {
    const int cls = (int)(i % a.num_class);
    const long nr = i / a.num_class;
    const int n = (int)(nr / a.R);
    const float4 b = reinterpret_cast<const float4*>(a.rois)[nr];
    const int decode_cls = a.class_agnostic ? 1 : cls;
    const float4 d = *reinterpret_cast<const float4*>(a.deltas + nr * 4 * a.K + decode_cls * 4);
    const float width = b.z - b.x + 1.0f, height = b.w - b.y + 1.0f;
    float px1, py1, px2, py2;
    if (!a.xyxy) {
      const float ctr_x = b.x + 0.5f * (width - 1.0f), ctr_y = b.y + 0.5f * (height - 1.0f);
      const float dx = d.x * a.stds[0] + a.means[0], dy = d.y * a.stds[1] + a.means[1];
      const float dw = d.z * a.stds[2] + a.means[2], dh = d.w * a.stds[3] + a.means[3];
      const float pred_ctr_x = dx * width + ctr_x, pred_ctr_y = dy * height + ctr_y;
      const float pred_w = (float)(exp((double)dw) * (double)width);
      const float pred_h = (float)(exp((double)dh) * (double)height);
      px1 = pred_ctr_x - 0.5f * (pred_w - 1.0f);
      py1 = pred_ctr_y - 0.5f * (pred_h - 1.0f);
      px2 = pred_ctr_x + 0.5f * (pred_w - 1.0f);
      py2 = pred_ctr_y + 0.5f * (pred_h - 1.0f);
    } else {
      px1 = b.x + (d.x * a.stds[0] + a.means[0]) * width;
      py1 = b.y + (d.y * a.stds[1] + a.means[1]) * height;
      px2 = b.z + (d.z * a.stds[2] + a.means[2]) * width;
      py2 = b.w + (d.w * a.stds[3] + a.means[3]) * height;
    }
    const float wm = a.im_info[n * 3 + 1] - 1.0f, hm = a.im_info[n * 3 + 0] - 1.0f;
    // std::max(std::min(x, m), 0)
    float4 o;
    o.x = fmaxr(fminr(px1, wm), 0.0f);
    o.y = fmaxr(fminr(py1, hm), 0.0f);
    o.z = fmaxr(fminr(px2, wm), 0.0f);
    o.w = fmaxr(fminr(py2, hm), 0.0f);
    reinterpret_cast<float4*>(a.out)[i] = o;
  }
}

struct FilterArgs {
  const float* bbox;
  const float* score;
  float* dets;
  int* counts;
  int B, R, K, Kb;
  float thr;
};

__global__ __launch_bounds__(256) void det_filter_kernel(FilterArgs a) {
  const int lane = threadIdx.x & (kWave - 1);
  const int prob = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave;
  if (prob >= a.B * a.K) return;
  const int n = prob / a.K, cid = prob % a.K;
  float* d = a.dets + (long)prob * a.R * 5;
  int m = 0;
  for (int r0 = 0; r0 < a.R; r0 += kWave) {
    const int r = r0 + lane;
    float s = 0.f;
    bool take = false;
    if (r < a.R) {
      s = a.score[((long)n * a.R + r) * a.K + cid];
      take = s > a.thr;
    }
    const unsigned long long bal = __ballot(take);
    if (take) {
      const int pos = m + __popcll(bal & ((1ull << lane) - 1));
      const float4 b = *reinterpret_cast<const float4*>(
          a.bbox + ((long)n * a.R + r) * 4 * a.Kb + (a.Kb == 1 ? 0 : cid * 4));
      float* o = d + (long)pos * 5;
      o[0] = b.x; o[1] = b.y; o[2] = b.z; o[3] = b.w; o[4] = s;
    }
    m += __popcll(bal);
  }
  if (lane == 0) a.counts[prob] = m;
}

}  // namespace sd

using namespace sd;

extern "C" int sd_decode_bbox(const float* rois, const float* bbox_pred, const float* im_info,
                              float* out, int B, int R, int K, const float* bbox_mean_host,
                              const float* bbox_std_host, int class_agnostic, int decode_xyxy,
                              void* stream) {
  SD_REQUIRE(B >= 0 && R >= 0 && K >= 1, "bad dimensions");
  SD_REQUIRE(bbox_mean_host && bbox_std_host, "bbox_mean / bbox_std null");
  SD_REQUIRE(!class_agnostic || K >= 2, "class_agnostic decoding reads the deltas of class 1");
  if ((long)B * R == 0) return SD_OK;
  SD_REQUIRE(rois && bbox_pred && im_info && out, "null tensor pointer");
  SD_REQUIRE((((uintptr_t)rois | (uintptr_t)bbox_pred | (uintptr_t)out) & 15) == 0,
             "rois / bbox_pred / out must be 16-byte aligned");
  DecodeArgs a{rois, bbox_pred, im_info, out, B, R, K, class_agnostic ? 1 : K, class_agnostic,
               decode_xyxy, {}, {}};
  for (int i = 0; i < 4; ++i) {
    a.means[i] = bbox_mean_host[i];
    a.stds[i] = bbox_std_host[i];
  }
  const long count = (long)B * R * a.num_class;
  const int grid = (int)((count + 255) / 256 < kNumCU * 16 ? (count + 255) / 256 : kNumCU * 16);
  hipLaunchKernelGGL(decode_bbox_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  SD_LAUNCH_CHECK();
  return SD_OK;
}

extern "C" int sd_det_filter(const float* bbox, const float* cls_score, int B, int R, int K,
                             int bbox_classes, float min_det_score, float* dets, int32_t* counts,
                             void* stream) {
  SD_REQUIRE(B >= 0 && R >= 0 && K >= 0, "negative dimension");
  SD_REQUIRE(bbox_classes == 1 || bbox_classes == K, "bbox must hold 1 or K boxes per roi");
  if ((long)B * K == 0) return SD_OK;
  SD_REQUIRE(counts, "counts is null");
  if (R == 0) {
    SD_HIP_CHECK(hipMemsetAsync(counts, 0, sizeof(int) * (size_t)B * K, (hipStream_t)stream));
    return SD_OK;
  }
  SD_REQUIRE(bbox && cls_score && dets, "null tensor pointer");
  SD_REQUIRE(((uintptr_t)bbox & 15) == 0, "bbox must be 16-byte aligned");
  FilterArgs a{bbox, cls_score, dets, counts, B, R, K, bbox_classes, min_det_score};
  hipLaunchKernelGGL(det_filter_kernel, dim3((B * K + 3) / 4), dim3(256), 0, (hipStream_t)stream, a);
  SD_LAUNCH_CHECK();
  return SD_OK;
}
