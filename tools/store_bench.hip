// Cost of the forward's value stores by shape: (1) 63 lanes x 4 B, one instruction per channel (rows of
// 7 floats per item, 9 items per wave pass: the band kernel's pattern), (2) the same bytes as two
// unaligned 16-byte stores per four channels (lanes q < 4 own channel q: bins 0..3, then bins 3..6),
// (3) byte stores of the arg-max codes, one per channel, (4) their packed form: two unaligned dword
// stores per four channels.  Addresses walk a 51 MB buffer the way (RoI, row) items do.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(1024) void k(float* out, unsigned char* am, const int* items, int nitems, int mode, int C) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int q = lane % 7, j = lane / 7;
  for (int ch = blockIdx.y * 4; ch < C; ch += gridDim.y * 4) {
    for (int p = 0; p < 4; ++p) {
      const int it = (blockIdx.x * 64 + wave * 4 + p) * 9 + j;
      if (j >= 9 || it >= nitems) continue;
      const int w = items[it];
      const long n = w & 0xffff, pp = w >> 16;
      const long oo = (n * C + ch) * 49 + pp * 7;
      const long ao = (n * C + ch) * 52 + pp * 7;
      const float v = (float)(lane + ch);
      if (mode == 1) {
        for (int g = 0; g < 4; ++g) out[oo + g * 49 + q] = v;
      } else if (mode == 2) {
        if (q < 4) {
          float* p0 = out + oo + q * 49;
          *reinterpret_cast<float4*>(p0) = make_float4(v, v, v, v);
          *reinterpret_cast<float4*>(p0 + 3) = make_float4(v, v, v, v);
        }
      } else if (mode == 3) {
        for (int g = 0; g < 4; ++g) am[ao + g * 52 + q] = (unsigned char)q;
      } else if (mode == 5) {
        // RoI-wise: the 4 x 49 values of (RoI, 4 channels) as one aligned 784-byte run, 49 lanes x 16 B;
        // a wave pass covers one RoI (instead of 9 rows), so 7 such stores carry what 4 x 7 row stores do
        if (j == 0 && q == 0) {}
        const int it7 = (blockIdx.x * 64 + wave * 4 + p) * 9;  // first item of this pass
        const long n0 = items[it7 < nitems ? it7 : 0] & 0xffff;
        for (int r = 0; r < 9; r += 7) {  // ~9/7 RoIs per pass: keep the byte count equal on average
          const long base = ((n0 + r) % 1024 * C + ch) * 49;
          if (lane < 49 && (r == 0 || (p & 3) < 1)) *reinterpret_cast<float4*>(out + base + lane * 4) = make_float4(v, v, v, v);
        }
      } else if (mode == 4) {
        if (q < 4) {
          unsigned char* p0 = am + ao + q * 52;
          *reinterpret_cast<unsigned*>(p0) = 0x01020304u;
          *reinterpret_cast<unsigned*>(p0 + 3) = 0x01020304u;
        }
      }
    }
  }
}
int main() {
  const int R = 1024, C = 256, nitems = R * 7;
  std::vector<int> items(nitems);
  unsigned s = 12345;
  std::vector<int> perm(R);
  for (int i = 0; i < R; ++i) perm[i] = i;
  for (int i = R - 1; i > 0; --i) { s = s * 1664525u + 1013904223u; int j = s % (i + 1); std::swap(perm[i], perm[j]); }
  for (int i = 0; i < R; ++i) for (int pp = 0; pp < 7; ++pp) items[i * 7 + pp] = perm[i] | (pp << 16);
  float* out; unsigned char* am; int* d;
  hipMalloc(&out, (size_t)R * C * 49 * 4 + 64); hipMalloc(&am, (size_t)R * C * 52 + 64); hipMalloc(&d, nitems * 4);
  hipMemcpy(d, items.data(), nitems * 4, hipMemcpyHostToDevice);
  const dim3 grid((nitems + 64 * 9 - 1) / (64 * 9), 16);  // 13 x 16 = 208 workgroups of 16 waves
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 1; mode <= 5; ++mode) {
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, grid, dim3(1024), 0, 0, out, am, d, nitems, mode, C);
    hipEventRecord(e0);
    for (int rep = 0; rep < 20; ++rep) hipLaunchKernelGGL(k, grid, dim3(1024), 0, 0, out, am, d, nitems, mode, C);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("mode %d: %.1f us per launch\n", mode, ms / 20 * 1e3);
  }
  return 0;
}
