// microbenchmark: the backward-plane scatter pattern (2x2 patch of adds per item into an LDS plane,
// item values streamed from global memory) with the three accumulator types
//   0: float compare-and-swap add   1: 64-bit fixed point ds_add_u64   2: 32-bit ds_add_u32
// build: hipcc --offload-arch=gfx950 -O3 -o tools/lds_scatter_bench tools/lds_scatter_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
template <int MODE, int T>
__global__ __launch_bounds__(T) void k(const float* __restrict__ val, const int* __restrict__ pos,
                                       float* out, int items, int W, int plane) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  long long* pq = reinterpret_cast<long long*>(smem);
  int* pi = reinterpret_cast<int*>(smem);
  const int words = MODE == 1 ? 2 * plane : plane;
  for (int i = threadIdx.x; i < words; i += T) smem[i] = 0.f;
  __syncthreads();
  const long base = (long)blockIdx.x * items;
  for (int it = threadIdx.x; it < items; it += T) {
    const float v = val[base + it];
    const int a = pos[base + it];
    const float w[4] = {v * 0.25f, v * 0.3f, v * 0.2f, v * 0.25f};
    const int o[4] = {a, a + 1, a + W, a + W + 1};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (MODE == 0) {
        int* ip = pi + o[j];
        int old = *ip;
        while (true) {
          const int assumed = old;
          old = atomicCAS(ip, assumed, __float_as_int(__int_as_float(assumed) + w[j]));
          if (old == assumed) break;
        }
      } else if (MODE == 1) {
        const long long q = (long long)ldexpf(w[j], 30);
        __hip_atomic_fetch_add(reinterpret_cast<unsigned long long*>(pq + o[j]), (unsigned long long)q,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      } else {
        const int q = (int)ldexpf(w[j], 16);
        __hip_atomic_fetch_add(reinterpret_cast<unsigned*>(pi + o[j]), (unsigned)q, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  }
  __syncthreads();
  float s = 0;
  for (int i = threadIdx.x; i < words; i += T) s += smem[i];
  if (s == 12345.f) out[0] = s;
}
template <int MODE, int T>
void run(const char* name, const float* val, const int* pos, float* out, int blocks, int items, int W,
         int plane, int clustered) {
  const size_t lds = (size_t)plane * (MODE == 1 ? 8 : 4);
  auto kern = k<MODE, T>;
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(T), lds, 0, val, pos, out, items, W, plane);
  hipEventRecord(a);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(T), lds, 0, val, pos, out, items, W, plane);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  const double adds = (double)blocks * items * 4;
  printf("%-18s T=%4d %s: %.3f ms  %.2f adds/clk/CU  (%.0f M adds)\n", name, T,
         clustered ? "clustered" : "random   ", ms, adds / (ms * 1e-3) / 2.4e9 / 256, adds / 1e6);
}
int main() {
  const int W = 336, rows = 25, plane = W * rows + W + 4, blocks = 6144, items = 2700;
  float* val; int* pos; float* out;
  const size_t n = (size_t)blocks * items;
  hipMalloc(&val, n * 4); hipMalloc(&pos, n * 4); hipMalloc(&out, 4);
  float* hv = (float*)malloc(n * 4); int* hp = (int*)malloc(n * 4);
  for (int clustered = 0; clustered < 2; ++clustered) {
    unsigned x = 12345u;
    for (size_t i = 0; i < n; ++i) {
      x = x * 1664525u + 1013904223u;
      hv[i] = (float)((x >> 8) & 0xffff) / 65536.f - 0.5f;
      x = x * 1664525u + 1013904223u;
      if (!clustered) hp[i] = (int)((x >> 8) % (unsigned)(W * (rows - 1) - 1));
      else {  // 49 consecutive items (one RoI) fall in a 12 x 12 pixel patch
        const size_t roi = i / 49;
        unsigned y = (unsigned)roi * 2654435761u;
        const int bx = (int)(y % (unsigned)(W - 14)), by = (int)((y >> 16) % (unsigned)(rows - 14));
        hp[i] = (by + (int)((x >> 8) % 12)) * W + bx + (int)((x >> 16) % 12);
      }
    }
    hipMemcpy(val, hv, n * 4, hipMemcpyHostToDevice); hipMemcpy(pos, hp, n * 4, hipMemcpyHostToDevice);
    run<0, 512>("float CAS", val, pos, out, blocks, items, W, plane, clustered);
    run<1, 512>("fixed64 ds_add_u64", val, pos, out, blocks, items, W, plane, clustered);
    run<1, 1024>("fixed64 ds_add_u64", val, pos, out, blocks, items, W, plane, clustered);
    run<2, 512>("fixed32 ds_add_u32", val, pos, out, blocks, items, W, plane, clustered);
  }
  return 0;
}
