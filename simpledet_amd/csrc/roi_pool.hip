// ROIPooling_v1 forward/backward for gfx950.
//   reference: operator_cxx/roi_pooling_v1.cu:48-113 (forward), :115-152 (backward scatter),
//              roi_pooling_v1.cc:39-126 (CPU forward, same result), roi_pooling_v1-inl.h:70-133
//              (pre-fill, req handling).
// forward: one wave per (roi, channel), lane = output bin, so the three tensors are written with
//   contiguous stores and the bins of one RoI share their cache lines; the RoI geometry is computed
//   once per wave in registers.  The kernel is bound by the instruction count of the per-bin scan
//   (~10 VALU per visited pixel; 435 M visits for 1024 RoIs x 1024 channels), not by memory: an
//   LDS-window variant (coalesced staging, several channels in flight, channel-major XCD order) was
//   measured at the same 0.7 ms and removed.
// backward (roi_pool_bwd_lds_kernel): workgroup = (image, channel, row band) with the band of dX
//   in LDS; the bins of the image's RoIs whose arg-max falls in the band are added with an LDS
//   compare-and-swap and the band is written once: no zero-fill pass, no global atomics
//   (1.71 -> 0.24 ms on the C4 shape).  roi_pool_bwd = 0 selects the reference structure
//   (zero-fill + global atomics), also the fallback when the RoI list does not fit in LDS.
#include "common.h"
#include "../../include/simpledet_ops.h"
#include <float.h>
#include <math.h>

namespace sd {

struct PoolArgs {
  const float* data;
  const float* rois;
  float* out;
  float* maxidx;
  int B, C, H, W, K, PH, PW;
  float scale;
};

__global__ __launch_bounds__(256) void roi_pool_fwd_kernel(PoolArgs a) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x / kWave;
  const int waves_per_block = blockDim.x / kWave;
  const int PP = a.PH * a.PW;
  const long nwork = (long)a.K * a.C;  // (roi, channel) pairs
  for (long wk = (long)blockIdx.x * waves_per_block + wave; wk < nwork;
       wk += (long)gridDim.x * waves_per_block) {
    const int n = (int)(wk / a.C), c = (int)(wk % a.C);
    const float* r = a.rois + (long)n * 5;
    const int roi_batch_ind = (int)r[0];
    // round(): half away from zero, on the float product (roi_pooling_v1.cu:70-73)
    const int roi_start_w = (int)roundf(r[1] * a.scale);
    const int roi_start_h = (int)roundf(r[2] * a.scale);
    const int roi_end_w = (int)roundf(r[3] * a.scale);
    const int roi_end_h = (int)roundf(r[4] * a.scale);
    const int roi_width = imaxr(roi_end_w - roi_start_w + 1, 1);
    const int roi_height = imaxr(roi_end_h - roi_start_h + 1, 1);
    const float bin_size_h = (float)roi_height / (float)a.PH;
    const float bin_size_w = (float)roi_width / (float)a.PW;
    const bool batch_ok = roi_batch_ind >= 0 && roi_batch_ind < a.B;
    const float* plane = a.data + ((long)(batch_ok ? roi_batch_ind : 0) * a.C + c) * a.H * a.W;
    const long obase = wk * PP;
    for (int bin = lane; bin < PP; bin += kWave) {
      const int ph = bin / a.PW, pw = bin % a.PW;
      int hstart = (int)floorf((float)ph * bin_size_h);
      int wstart = (int)floorf((float)pw * bin_size_w);
      int hend = (int)ceilf((float)(ph + 1) * bin_size_h);
      int wend = (int)ceilf((float)(pw + 1) * bin_size_w);
      hstart = iminr(imaxr(hstart + roi_start_h, 0), a.H);
      hend = iminr(imaxr(hend + roi_start_h, 0), a.H);
      wstart = iminr(imaxr(wstart + roi_start_w, 0), a.W);
      wend = iminr(imaxr(wend + roi_start_w, 0), a.W);
      const bool is_empty = (hend <= hstart) || (wend <= wstart) || !batch_ok;
      float maxval = is_empty ? 0.f : -FLT_MAX;
      int maxi = -1;
      if (!is_empty) {
        for (int h = hstart; h < hend; ++h) {
          const float* row = plane + (long)h * a.W;
          for (int w = wstart; w < wend; ++w) {
            const float v = row[w];
            if (v > maxval) {
              maxval = v;
              maxi = h * a.W + w;
            }
          }
        }
      }
      a.out[obase + bin] = maxval;
      a.maxidx[obase + bin] = (float)maxi;
    }
  }
}

struct PoolBwdArgs {
  const float* dy;
  const float* rois;
  const float* maxidx;
  float* dx;
  int B, C, H, W, K, PP;
};

__global__ __launch_bounds__(256) void roi_pool_bwd_kernel(PoolBwdArgs a) {
  const long count = (long)a.K * a.C * a.PP;
  for (long index = (long)blockIdx.x * blockDim.x + threadIdx.x; index < count;
       index += (long)gridDim.x * blockDim.x) {
    const int c = (int)((index / a.PP) % a.C);
    const int n = (int)(index / a.PP / a.C);
    const int argmax = (int)a.maxidx[index];
    if (argmax != -1) {
      const int b = (int)a.rois[(long)n * 5];
      if (b >= 0 && b < a.B && argmax >= 0 && argmax < a.H * a.W)
        atomicAdd(a.dx + ((long)b * a.C + c) * a.H * a.W + argmax, a.dy[index]);
    }
  }
}


// ------------------------------------------------------------------------------------------------
// LDS-plane backward
// ------------------------------------------------------------------------------------------------

// grid: x = channel, y = row band, z = image
__global__ __launch_bounds__(512) void roi_pool_bwd_lds_kernel(PoolBwdArgs a, int band_rows,
                                                               int req_add) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int T = 512;
  const int tid = threadIdx.x;
  const int c = blockIdx.x, b = blockIdx.z;
  const int row0 = blockIdx.y * band_rows, row1 = iminr(row0 + band_rows, a.H);
  const int band_elems = (row1 - row0) * a.W;
  const int plane_pad = (band_elems + 3) & ~3;
  float* plane = smem;
  int* list = reinterpret_cast<int*>(smem + plane_pad);
  int* nlist = list + a.K;
  for (int i = tid; i < plane_pad; i += T) plane[i] = 0.f;
  if (tid == 0) *nlist = 0;
  __syncthreads();
  for (int n = tid; n < a.K; n += T)
    if ((int)a.rois[(long)n * 5] == b) list[atomicAdd(nlist, 1)] = n;
  __syncthreads();
  const int nitems = *nlist * a.PP;
  const int lo = row0 * a.W, hi = row1 * a.W;
  for (int it = tid; it < nitems; it += T) {
    const int n = list[it / a.PP], bin = it % a.PP;
    const long idx = ((long)n * a.C + c) * a.PP + bin;
    const int argmax = (int)a.maxidx[idx];
    if (argmax >= lo && argmax < hi) lds_add_cas(plane + (argmax - lo), a.dy[idx]);
  }
  __syncthreads();
  float* dst = a.dx + (((long)b * a.C + c) * a.H + row0) * a.W;
  for (int i = tid; i < band_elems; i += T) dst[i] = req_add ? dst[i] + plane[i] : plane[i];
}

}  // namespace sd

using namespace sd;

extern "C" int sd_roi_pool_v1_fwd(const float* data, const float* rois, float* out, float* maxidx,
                                  int B, int C, int H, int W, int K, int pooled_h, int pooled_w,
                                  float spatial_scale, void* stream) {
  SD_REQUIRE(B >= 0 && C >= 0 && K >= 0 && H > 0 && W > 0, "bad dimensions");
  SD_REQUIRE(pooled_h > 0 && pooled_w > 0, "pooled_size must be nonzero");
  SD_REQUIRE((long)K * C * pooled_h * pooled_w < (1L << 31), "output has >= 2^31 elements");
  SD_REQUIRE((long)H * W < (1L << 24), "plane too large for a float argmax index");
  if ((long)K * C == 0) return SD_OK;
  SD_REQUIRE(data && rois && out && maxidx, "null tensor pointer");
  PoolArgs a{data, rois, out, maxidx, B, C, H, W, K, pooled_h, pooled_w, spatial_scale};
  const long nwork = (long)K * C;
  const int grid = (int)((nwork + 3) / 4 < kNumCU * 16 ? (nwork + 3) / 4 : kNumCU * 16);
  hipLaunchKernelGGL(roi_pool_fwd_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
  SD_LAUNCH_CHECK();
  return SD_OK;
}

extern "C" int sd_roi_pool_v1_bwd(const float* out_grad, const float* rois, const float* maxidx,
                                  float* d_data, float* d_rois, int req_data, int req_rois, int B,
                                  int C, int H, int W, int K, int pooled_h, int pooled_w,
                                  float spatial_scale, void* stream) {
  (void)spatial_scale;
  SD_REQUIRE(B >= 0 && C >= 0 && K >= 0 && H > 0 && W > 0, "bad dimensions");
  SD_REQUIRE(pooled_h > 0 && pooled_w > 0, "pooled_size must be nonzero");
  SD_REQUIRE(req_data == SD_REQ_NULL || req_data == SD_REQ_WRITE || req_data == SD_REQ_ADD,
             "ROIPooling: Backward doesn't support kWriteInplace.");
  SD_REQUIRE(req_rois == SD_REQ_NULL || req_rois == SD_REQ_WRITE || req_rois == SD_REQ_ADD,
             "ROIPooling: Backward doesn't support kWriteInplace.");
  hipStream_t st = (hipStream_t)stream;
  const size_t dx_bytes = (size_t)B * C * H * W * sizeof(float);
  if (req_data != SD_REQ_NULL && dx_bytes) {
    SD_REQUIRE(d_data, "d_data is null");
    const long count = (long)K * C * pooled_h * pooled_w;
    SD_REQUIRE(count == 0 || (out_grad && rois && maxidx), "null tensor pointer");
    PoolBwdArgs a{out_grad, rois, maxidx, d_data, B, C, H, W, K, pooled_h * pooled_w};
    // row bands of at most 36 KB so that four workgroups share a CU
    const long budget = 36 * 1024;
    int nb = (int)(((long)H * W * 4 + budget - 1) / budget);
    if (nb < 1) nb = 1;
    int rows = (H + nb - 1) / nb;
    nb = (H + rows - 1) / rows;
    const size_t lds = (size_t)((((long)rows * W + 3) & ~3L) * 4) + (size_t)(K + 4) * 4;
    if (lds <= 150 * 1024 && C <= 65535 * 32 && nb <= 65535 && B <= 65535 &&
        tuning("roi_pool_bwd", 1) == 1) {
      if (lds > 64 * 1024)
        SD_HIP_CHECK(hipFuncSetAttribute((const void*)roi_pool_bwd_lds_kernel,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      hipLaunchKernelGGL(roi_pool_bwd_lds_kernel, dim3(C, nb, B), dim3(512), lds, st, a, rows,
                         req_data == SD_REQ_ADD ? 1 : 0);
      SD_LAUNCH_CHECK();
    } else {
      if (req_data == SD_REQ_WRITE) SD_HIP_CHECK(hipMemsetAsync(d_data, 0, dx_bytes, st));
      if (count) {
        const int grid = (int)((count + 255) / 256 < kNumCU * 32 ? (count + 255) / 256 : kNumCU * 32);
        hipLaunchKernelGGL(roi_pool_bwd_kernel, dim3(grid), dim3(256), 0, st, a);
        SD_LAUNCH_CHECK();
      }
    }
  }
  if (req_rois == SD_REQ_WRITE && K > 0) {  // roi_pooling_v1-inl.h:130-132
    SD_REQUIRE(d_rois, "d_rois is null but req_rois == write");
    SD_HIP_CHECK(hipMemsetAsync(d_rois, 0, (size_t)K * 5 * sizeof(float), st));
  }
  return SD_OK;
}
