// Scratch microbenchmark (GPU box): what does timing a chain of kernels cost?
//   (1) plain launches, (2) hipEventRecord between the launches (what torch.cuda.Event.record does),
//   (3) hipExtLaunchKernelGGL with start / stop events (the dispatch's own timestamps, no marker packet).
// build: hipcc --offload-arch=gfx950 -O3 tools/ext_event_bench.hip -o tools/ext_event_bench
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void copy_kernel(const float4* __restrict__ a, float4* __restrict__ b, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) b[i] = a[i];
}
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(err_)); return 1; } } while (0)

int main() {
  const long n = 160L << 20 >> 4;   // 160 MB read + 160 MB written per launch
  float4 *a, *b;
  CK(hipMalloc(&a, n * 16)); CK(hipMalloc(&b, n * 16));
  CK(hipMemset(a, 1, n * 16));
  hipStream_t st; CK(hipStreamCreate(&st));
  const int steps = 200, per = 3;
  auto launch = [&]() { hipLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, st, a, b, n); };
  for (int i = 0; i < 50; ++i) launch();
  CK(hipStreamSynchronize(st));
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](auto t0, auto t1) { return std::chrono::duration<double, std::milli>(t1 - t0).count(); };
  for (int rep = 0; rep < 2; ++rep) {
    auto t0 = now();
    for (int s = 0; s < steps; ++s) for (int k = 0; k < per; ++k) launch();
    CK(hipStreamSynchronize(st));
    const double plain = ms(t0, now()) / steps;
    std::vector<hipEvent_t> ev(steps * (per + 1));
    for (auto& e : ev) CK(hipEventCreate(&e));
    t0 = now();
    for (int s = 0; s < steps; ++s) {
      CK(hipEventRecord(ev[s * (per + 1)], st));
      for (int k = 0; k < per; ++k) { launch(); CK(hipEventRecord(ev[s * (per + 1) + k + 1], st)); }
    }
    CK(hipStreamSynchronize(st));
    const double rec = ms(t0, now()) / steps;
    float sum = 0, t;
    for (int s = 0; s < steps; ++s) for (int k = 0; k < per; ++k) { CK(hipEventElapsedTime(&t, ev[s * (per + 1) + k], ev[s * (per + 1) + k + 1])); sum += t; }
    const double rec_kernel = sum / (steps * per);
    std::vector<hipEvent_t> e0(steps * per), e1(steps * per);
    for (auto& e : e0) CK(hipEventCreate(&e));
    for (auto& e : e1) CK(hipEventCreate(&e));
    t0 = now();
    for (int s = 0; s < steps; ++s)
      for (int k = 0; k < per; ++k)
        hipExtLaunchKernelGGL(copy_kernel, dim3(2048), dim3(256), 0, st, e0[s * per + k], e1[s * per + k], 0, a, b, n);
    CK(hipStreamSynchronize(st));
    const double ext = ms(t0, now()) / steps;
    sum = 0;
    float span = 0;
    for (int i = 0; i < steps * per; ++i) { CK(hipEventElapsedTime(&t, e0[i], e1[i])); sum += t; }
    for (int s = 0; s < steps; ++s) { CK(hipEventElapsedTime(&t, e0[s * per], e1[s * per + per - 1])); span += t; }
    printf("per step of %d launches: plain %.4f ms | hipEventRecord between %.4f ms (event-timed kernel %.4f ms) | "
           "hipExtLaunch events %.4f ms (kernel %.4f ms, first-start..last-stop %.4f ms)\n",
           per, plain, rec, rec_kernel, ext, sum / (steps * per), span / steps);
  }
  return 0;
}
