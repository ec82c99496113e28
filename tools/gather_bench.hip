// Microbenchmark (scratch, not product code): throughput of the RoIAlign forward's tile-fill access
// patterns on MI355X, without any compute.  Each wave repeatedly fetches one "tile" (the taps of one
// RoI on one channel plane) from a 2 x 256 x 200 x 334 fp32 feature map at pseudo-random positions.
//   pattern 0: 7 x dwordx2 gathers, 56 lanes  (row slot r0, column pair jp)        3136 B / tile
//   pattern 1: 4 x dwordx4 gathers, 196 lanes (row slot, bin column), 4-float window 3136 B / tile
//   pattern 2: 4 x dwordx4 dense rows, 28 rows x 32 floats                          3584 B / tile
//   pattern 3: 14 x dword dense rows, 28 rows x 32 floats                           3584 B / tile
// build: hipcc --offload-arch=gfx950 -O3 -o gather_bench gather_bench.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

constexpr int C = 256, B = 2;
__constant__ int gH, gW;
#define H gH
#define W gW

struct __attribute__((packed, aligned(4))) F2u { float x, y; };
struct __attribute__((packed, aligned(4))) F4u { float x, y, z, w; };

__device__ __forceinline__ unsigned hash(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

template <int PATTERN, int DEPTH, bool TO_LDS>
__global__ __launch_bounds__(512) void gather_kernel(const float* __restrict__ feat, float* out,
                                                     int tiles_per_wave, int nslice) {
  __shared__ __attribute__((aligned(16))) float lds[8 * 1024];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int grp = blockIdx.x / nslice, slice = blockIdx.x % nslice;
  const int nch = C / nslice;
  float* tile = lds + wave * 1024;
  float acc = 0.f;
  for (int t0 = 0; t0 < tiles_per_wave; t0 += DEPTH) {
    float v[DEPTH][16];
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      const int t = t0 + d;
      // tile -> (roi position from the RoI group, channel of this wave)
      const unsigned h = hash((unsigned)(grp * 4 + (t / (nch / 8))));
      const int img = h & 1;
      const int row0 = (h >> 1) % (H > 30 ? H - 30 : 1), col0 = (h >> 12) % (W > 36 ? W - 36 : 1);
      const int c = slice * nch + wave + 8 * (t % (nch / 8));
      const float* pl = feat + ((long)(img * C + c) * H) * W;
      if (PATTERN == 0) {
        const int r0 = lane / 14, jp = lane % 14;
#pragma unroll
        for (int it = 0; it < 7; ++it) {
          const int off = ((row0 + 4 * it + r0) % H) * W + col0 + 2 * jp;
          const F2u x = *reinterpret_cast<const F2u*>(pl + (lane < 56 ? off : 0));
          v[d][2 * it] = x.x; v[d][2 * it + 1] = x.y;
        }
      } else if (PATTERN == 1) {
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int L = it * 64 + lane, rr = L / 7, q = L % 7;
          const int off = ((row0 + rr) % H) * W + col0 + 4 * q;
          const F4u x = *reinterpret_cast<const F4u*>(pl + (L < 196 ? off : 0));
          v[d][4 * it] = x.x; v[d][4 * it + 1] = x.y; v[d][4 * it + 2] = x.z; v[d][4 * it + 3] = x.w;
        }
      } else if (PATTERN == 2) {
        const int ca = col0 & ~3;
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int r = it * 8 + (lane >> 3);
          const int off = ((row0 + (r < 28 ? r : 0)) % H) * W + ca + 4 * (lane & 7);
          const F4u x = *reinterpret_cast<const F4u*>(pl + off);
          v[d][4 * it] = x.x; v[d][4 * it + 1] = x.y; v[d][4 * it + 2] = x.z; v[d][4 * it + 3] = x.w;
        }
      } else {
        const int ca = col0 & ~3;
#pragma unroll
        for (int it = 0; it < 14; ++it) {
          const int r = it * 2 + (lane >> 5);
          v[d][it] = pl[((row0 + r) % H) * W + ca + (lane & 31)];
        }
        v[d][14] = v[d][15] = 0.f;
      }
    }
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) {
      if (TO_LDS) {
#pragma unroll
        for (int e = 0; e < 16; e += 4)
          *reinterpret_cast<float4*>(tile + (e / 4) * 256 + lane * 4) =
              make_float4(v[d][e], v[d][e + 1], v[d][e + 2], v[d][e + 3]);
        const float4 r = *reinterpret_cast<const float4*>(tile + ((lane * 5) & 63) * 4);
        acc += r.x + r.y + r.z + r.w;
      } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) acc += v[d][e];
      }
    }
  }
  if (acc == 12345.678f) out[blockIdx.x * 512 + threadIdx.x] = acc;  // keep the loads alive
}

template <int P, int D, bool L>
static void run(const char* name, const float* feat, float* out, int nslice, double bytes_per_tile) {
  const int groups = 256;                     // RoI groups (as the product kernel: 1024 RoIs / 4)
  const int tiles_per_wave = 4 * (C / nslice) / 8;  // 4 RoIs x channels of the slice / 8 waves
  const int grid = groups * nslice;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i)
    hipLaunchKernelGGL((gather_kernel<P, D, L>), dim3(grid), dim3(512), 0, 0, feat, out, tiles_per_wave, nslice);
  hipEventRecord(e0);
  const int iters = 20;
  for (int i = 0; i < iters; ++i)
    hipLaunchKernelGGL((gather_kernel<P, D, L>), dim3(grid), dim3(512), 0, 0, feat, out, tiles_per_wave, nslice);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  ms /= iters;
  const double tiles = (double)grid * 8 * tiles_per_wave;
  printf("%-46s %8.1f us  %7.2f TB/s useful  (%.0f tiles)\n", name, ms * 1e3,
         tiles * bytes_per_tile / (ms * 1e-3) / 1e12, tiles);
}

int main() {
  for (int cfg = 0; cfg < 3; ++cfg) {
    const int hh = cfg == 0 ? 200 : (cfg == 1 ? 50 : 25), ww = cfg == 0 ? 334 : (cfg == 1 ? 84 : 42);
    hipMemcpyToSymbol(HIP_SYMBOL(gH), &hh, sizeof(int));
    hipMemcpyToSymbol(HIP_SYMBOL(gW), &ww, sizeof(int));
    const size_t n = (size_t)B * C * hh * ww;
    float* feat; float* out;
    hipMalloc(&feat, n * 4 + 4096); hipMalloc(&out, 1 << 24);
    hipMemset(feat, 0, n * 4 + 4096);
    printf("== feature map 2 x 256 x %d x %d (%.1f MB)\n", hh, ww, n * 4 / 1e6);
    run<0, 1, false>("pairs 7x dwordx2 (56 lanes), depth 1", feat, out, 8, 3136);
    run<1, 1, false>("window 4x dwordx4 (196 lanes), depth 1", feat, out, 8, 3136);
    run<2, 1, false>("dense rows 4x dwordx4, depth 1", feat, out, 8, 3584);
    run<2, 4, false>("dense rows 4x dwordx4, depth 4", feat, out, 8, 3584);
    run<3, 1, false>("dense rows 14x dword, depth 1", feat, out, 8, 3584);
    run<0, 1, true>("pairs + LDS round trip, depth 1", feat, out, 8, 3136);
    hipFree(feat); hipFree(out);
  }
  return 0;
}
