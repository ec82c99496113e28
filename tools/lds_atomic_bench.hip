// microbenchmark: LDS atomic throughput on gfx950 (float vs int32 vs int64 vs plain RMW)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, int span) {
  __shared__ __attribute__((aligned(16))) unsigned long long buf64[4096];
  float* bf = reinterpret_cast<float*>(buf64);
  unsigned* bu = reinterpret_cast<unsigned*>(buf64);
  for (int i = threadIdx.x; i < 8192; i += 256) bf[i] = 0.f;
  __syncthreads();
  unsigned x = threadIdx.x * 2654435761u + blockIdx.x;
  for (int i = 0; i < iters; ++i) {
    x = x * 1664525u + 1013904223u;
    int idx = (x >> 8) & (span - 1);  // span is a power of two: keep the loop LDS- not VALU-bound
    if (MODE == 0) __hip_atomic_fetch_add(&bf[idx], 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (MODE == 1) __hip_atomic_fetch_add(&bu[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (MODE == 2) __hip_atomic_fetch_add(&buf64[idx >> 1], 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (MODE == 3) bf[idx] += 1.0f;
    else if (MODE == 4) __hip_atomic_fetch_max(&bu[idx], x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    else if (MODE == 5) {  // the float CAS add the RoIAlign / DCN backward planes use
      int* ip = reinterpret_cast<int*>(&bf[idx]);
      int old = *ip;
      while (true) {
        const int assumed = old;
        old = atomicCAS(ip, assumed, __float_as_int(__int_as_float(assumed) + 1.0f));
        if (old == assumed) break;
      }
    } else if (MODE == 6) {  // 64-bit fixed point: float -> int64 conversion + ds_add_u64
      const float v = __int_as_float(0x3f800000 | (x & 0x7fffff));
      __hip_atomic_fetch_add(&buf64[idx >> 1], (unsigned long long)(long long)(v * 1048576.f),
                             __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if (MODE == 7) {  // returning integer add (latency exposed)
      x += __hip_atomic_fetch_add(&bu[idx], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
  }
  __syncthreads();
  float s = 0;
  for (int i = threadIdx.x; i < 8192; i += 256) s += bf[i];
  if (s == 12345.f) out[0] = s;
}
template <int MODE>
void run(const char* name, int span) {
  float* d; hipMalloc(&d, 4);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int iters = 2000, blocks = 256 * 4;
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, span);
  hipEventRecord(a);
  hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters, span);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  double ops = (double)blocks * 256 * iters;
  printf("%-14s span %5d: %.3f ms  -> %.2f lane-ops/clk/CU (2.4GHz, 256 CU)\n", name, span, ms,
         ops / (ms * 1e-3) / 2.4e9 / 256);
  hipFree(d);
}
int main() {
  for (int span : {8192, 1024, 64}) {
    run<0>("ds_add_f32", span);
    run<1>("ds_add_u32", span);
    run<2>("ds_add_u64", span);
    run<3>("plain rmw f32", span);
    run<4>("ds_max_u32", span);
    run<5>("cas add f32", span);
    run<6>("cvt + add_u64", span);
    run<7>("ds_add_rtn_u32", span);
  }
  return 0;
}
