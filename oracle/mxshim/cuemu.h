// cuemu.h -- run the reference's CUDA translation units (*.cu) on the host CPU.
//
// TEST INFRASTRUCTURE ONLY.  The reference's GPU operators are the specification for several ops
// (_contrib_NMS and Proposal_v3 CPU paths differ from / are broken against
// the GPU ones) and nvcc / an NVIDIA GPU do not exist here.  oracle/build_ref.py feeds every
// `kernel<<<grid, block, ...>>>(args)` of a .cu through one textual rewrite,
//     CUEMU_LAUNCH(kernel, (grid, block, ...), (args))
// and compiles the otherwise unmodified file with g++ against this header.  A launch executes the
// kernel body once per (block, thread) with blockIdx/threadIdx set:
//   * kernels without __syncthreads(): a plain serial loop (so atomicAdd is a plain +=, and the
//     accumulation order is the thread-index order);
//   * kernels that call __syncthreads() (listed by build_ref.py in CUEMU_BARRIER_KERNELS):
//     blockDim real threads per block with a barrier; blocks run one after another, so a
//     `__shared__` array can be an ordinary static.
// CUDA's device math functions map to glibc's (floor/ceil/round/max/min are exact either way;
// expf/logf may differ from NVIDIA's in the last place: see DESIGN.md section 2).
#ifndef ORACLE_CUEMU_H_
#define ORACLE_CUEMU_H_

#include <math.h>
#include <stdint.h>
#include <string.h>
#include <stdlib.h>
#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__
#define __shared__ static
#define __CUDACC__ 1

struct uint3 { unsigned x, y, z; };
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}  // NOLINT
};

namespace cuemu {
struct Barrier {
  std::mutex m;
  std::condition_variable cv;
  unsigned n, waiting, gen;
  explicit Barrier(unsigned n_) : n(n_), waiting(0), gen(0) {}
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    const unsigned g = gen;
    if (++waiting == n) { waiting = 0; ++gen; cv.notify_all(); }
    else cv.wait(lk, [&] { return g != gen; });
  }
};
struct State {
  uint3 threadIdx, blockIdx;
  dim3 blockDim, gridDim;
  Barrier *barrier;
};
inline State &state() {
  static thread_local State s;
  return s;
}
inline bool needs_barrier(const char *kernel) {
#ifdef CUEMU_BARRIER_KERNELS
  const std::string list = "," CUEMU_BARRIER_KERNELS ",";
  std::string k = std::string(",") + kernel;
  const size_t lt = k.find('<');  // "kern<float>" -> "kern"
  if (lt != std::string::npos) k.erase(lt);
  return list.find(k + ",") != std::string::npos;
#else
  (void)kernel;
  return false;
#endif
}
struct LaunchCfg {
  dim3 grid, block;
  LaunchCfg(dim3 g, dim3 b, size_t = 0, void * = NULL) : grid(g), block(b) {}
};
template <typename F>
inline void launch(const char *kernel, const LaunchCfg &cfg, F body) {
  const dim3 g = cfg.grid, b = cfg.block;
  const unsigned nthreads = b.x * b.y * b.z;
  if (!needs_barrier(kernel)) {
    State &s = state();
    s.blockDim = b; s.gridDim = g; s.barrier = NULL;
    for (unsigned bz = 0; bz < g.z; ++bz) for (unsigned by = 0; by < g.y; ++by)
    for (unsigned bx = 0; bx < g.x; ++bx) {
      s.blockIdx.x = bx; s.blockIdx.y = by; s.blockIdx.z = bz;
      for (unsigned tz = 0; tz < b.z; ++tz) for (unsigned ty = 0; ty < b.y; ++ty)
      for (unsigned tx = 0; tx < b.x; ++tx) {
        s.threadIdx.x = tx; s.threadIdx.y = ty; s.threadIdx.z = tz;
        body();
      }
    }
    return;
  }
  // one OS thread per CUDA thread of a block; all of them walk the blocks in lock step
  Barrier bar(nthreads);
  std::vector<std::thread> pool;
  for (unsigned t = 0; t < nthreads; ++t) {
    pool.push_back(std::thread([&, t]() {
      State &s = state();
      s.blockDim = b; s.gridDim = g; s.barrier = &bar;
      s.threadIdx.x = t % b.x; s.threadIdx.y = (t / b.x) % b.y; s.threadIdx.z = t / (b.x * b.y);
      for (unsigned bz = 0; bz < g.z; ++bz) for (unsigned by = 0; by < g.y; ++by)
      for (unsigned bx = 0; bx < g.x; ++bx) {
        s.blockIdx.x = bx; s.blockIdx.y = by; s.blockIdx.z = bz;
        body();
        bar.wait();  // a block's shared memory is reused by the next block
      }
    }));
  }
  for (size_t t = 0; t < pool.size(); ++t) pool[t].join();
}
}  // namespace cuemu

#define threadIdx (::cuemu::state().threadIdx)
#define blockIdx (::cuemu::state().blockIdx)
#define blockDim (::cuemu::state().blockDim)
#define gridDim (::cuemu::state().gridDim)
inline void __syncthreads() {
  if (::cuemu::state().barrier) ::cuemu::state().barrier->wait();
}

#define CUEMU_UNPAREN(...) __VA_ARGS__
#define CUEMU_LAUNCH(kernel, cfg, args) \
  ::cuemu::launch(#kernel, ::cuemu::LaunchCfg(CUEMU_UNPAREN cfg), [&]() { kernel args; })

// ---- runtime API subset
typedef int cudaError_t;
static const cudaError_t cudaSuccess = 0;
enum cudaMemcpyKind { cudaMemcpyHostToHost, cudaMemcpyHostToDevice, cudaMemcpyDeviceToHost,
                      cudaMemcpyDeviceToDevice, cudaMemcpyDefault };
typedef void *cudaStream_t;
inline const char *cudaGetErrorString(cudaError_t) { return "cuemu: no error"; }
inline cudaError_t cudaPeekAtLastError() { return cudaSuccess; }
inline cudaError_t cudaGetLastError() { return cudaSuccess; }
inline cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
inline cudaError_t cudaStreamSynchronize(cudaStream_t) { return cudaSuccess; }
inline cudaError_t cudaMemcpy(void *dst, const void *src, size_t n, cudaMemcpyKind) {
  memcpy(dst, src, n);
  return cudaSuccess;
}
inline cudaError_t cudaMemcpyAsync(void *dst, const void *src, size_t n, cudaMemcpyKind, cudaStream_t = NULL) {
  memcpy(dst, src, n);
  return cudaSuccess;
}
inline cudaError_t cudaMemset(void *dst, int v, size_t n) { memset(dst, v, n); return cudaSuccess; }
template <typename T>
inline cudaError_t cudaMalloc(T **p, size_t n) { *p = static_cast<T *>(malloc(n)); return cudaSuccess; }
inline cudaError_t cudaFree(void *p) { free(p); return cudaSuccess; }

// serial execution: an atomic add is an add (returns the old value like CUDA's)
template <typename T>
inline T atomicAdd(T *addr, T v) { T old = *addr; *addr = old + v; return old; }

// CUDA's global-namespace min/max overloads
inline int max(int a, int b) { return a > b ? a : b; }
inline int min(int a, int b) { return a < b ? a : b; }
inline unsigned max(unsigned a, unsigned b) { return a > b ? a : b; }
inline unsigned min(unsigned a, unsigned b) { return a < b ? a : b; }
inline long long max(long long a, long long b) { return a > b ? a : b; }
inline long long min(long long a, long long b) { return a < b ? a : b; }
inline float max(float a, float b) { return fmaxf(a, b); }
inline float min(float a, float b) { return fminf(a, b); }
inline double max(double a, double b) { return fmax(a, b); }
inline double min(double a, double b) { return fmin(a, b); }
inline float max(float a, int b) { return fmaxf(a, static_cast<float>(b)); }
inline float max(int a, float b) { return fmaxf(static_cast<float>(a), b); }
inline float min(float a, int b) { return fminf(a, static_cast<float>(b)); }
inline float min(int a, float b) { return fminf(static_cast<float>(a), b); }
inline double max(double a, float b) { return fmax(a, static_cast<double>(b)); }
inline double max(float a, double b) { return fmax(static_cast<double>(a), b); }
inline double min(double a, float b) { return fmin(a, static_cast<double>(b)); }
inline double min(float a, double b) { return fmin(static_cast<double>(a), b); }

// ---- thrust subset (stable_sort_by_key with a comparator, device execution policy)
namespace thrust {
struct device_t {};
static const device_t device = device_t();
template <typename T> struct greater { bool operator()(const T &a, const T &b) const { return a > b; } };
template <typename T> struct less { bool operator()(const T &a, const T &b) const { return a < b; } };
template <typename K, typename V, typename Cmp>
inline void stable_sort_by_key(const device_t &, K *kb, K *ke, V *vb, Cmp cmp) {
  const size_t n = static_cast<size_t>(ke - kb);
  std::vector<size_t> idx(n);
  for (size_t i = 0; i < n; ++i) idx[i] = i;
  std::stable_sort(idx.begin(), idx.end(), [&](size_t a, size_t b) { return cmp(kb[a], kb[b]); });
  std::vector<K> k2(n);
  std::vector<V> v2(n);
  for (size_t i = 0; i < n; ++i) { k2[i] = kb[idx[i]]; v2[i] = vb[idx[i]]; }
  for (size_t i = 0; i < n; ++i) { kb[i] = k2[i]; vb[i] = v2[i]; }
}
template <typename K, typename V, typename Cmp>
inline void stable_sort_by_key(K *kb, K *ke, V *vb, Cmp cmp) {
  stable_sort_by_key(device, kb, ke, vb, cmp);
}
}  // namespace thrust

// ---- mshadow::cuda launch constants (mshadow/cuda/tensor_gpu-inl.cuh)
namespace mshadow {
namespace cuda {
static const int kBaseThreadBits = 8;
static const int kBaseThreadNum = 1 << kBaseThreadBits;
static const int kMaxThreadsPerBlock = 1024;
static const int kMaxGridNum = 65535;
static const int kMaxGridDim = 65535;
static const int kBaseGridNum = 1024;
inline void CheckLaunchParam(dim3 dimGrid, dim3 dimBlock, const char *estr = "") {
  if (dimBlock.x * dimBlock.y * dimBlock.z > static_cast<unsigned>(kMaxThreadsPerBlock) ||
      dimGrid.x > 65535 || dimGrid.y > 65535) {
    throw std::runtime_error(std::string("too large launch parameter: ") + estr);
  }
}
}  // namespace cuda
}  // namespace mshadow

#endif  // ORACLE_CUEMU_H_
