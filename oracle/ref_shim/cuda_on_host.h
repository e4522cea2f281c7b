// TEST INFRASTRUCTURE (oracle/_ref): a minimal CUDA execution model on the host.
//
// Purpose: compile the reference's OWN rasterizer sources -- forward.cu, backward.cu, rasterizer_impl.cu under
// /root/reference/submodules/diff-gaussian-rasterization/cuda_rasterizer, read where they lie, never copied -- with g++ and run
// them on the CPU, so that the hand-written oracle (oracle/radegs_oracle.cpp) and, through committed golden vectors, the HIP
// path can be compared with what the reference's code computes.  Nothing of the reference is restated here: this file only
// supplies what nvcc + the CUDA runtime + CUB supply.
//
//   * vector types (float2/3/4, uint2, dim3 ...), the __device__/__global__/__shared__ decorations (empty / thread_local static);
//   * the global-namespace min/max overload set of the CUDA math headers (incl. the mixed (unsigned,int) and (double,float)
//     forms the reference relies on: auxiliary.h:62-72, forward.cu:119-120);
//   * kernel launches: the build recipe (oracle/build_ref.py) rewrites the token sequence  k<<<g, b>>>(args)  to
//     k % cuda_on_host::cfg(g, b)(args)  on the fly -- the one edit g++ cannot do without -- and operator% below runs the grid;
//   * thread blocks as cooperative FIBERS (ucontext): every CUDA thread of a block is a fiber; block.sync() and
//     __syncthreads_count() yield to a round-robin scheduler, one pass over the fibers = one barrier phase.  __shared__ arrays are
//     thread_local statics of the OS thread that runs the block.  Blocks may run on several OS threads (OpenMP); with one thread
//     (the default) float atomics are applied in a fixed order and the backward is deterministic;
//   * cooperative_groups::this_grid()/this_thread_block(), atomicAdd(float*), cudaMemcpy/cudaMemset, and the two CUB entry points
//     the reference calls (DeviceScan::InclusiveSum, DeviceRadixSort::SortPairs = a STABLE sort on key bits [begin,end), which is
//     the documented contract of CUB's LSD radix sort; CUB itself is a CUDA-toolkit component absent from this image);
//   * `exp` on floats goes through a hook so that a run can use glibc's expf or the oracle's specified exponential
//     (CUDA's expf cannot be reproduced off-device either way).
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <iostream>
#include <math.h>
#include <numeric>
#include <stdexcept>
#include <stdio.h>
#include <tuple>
#include <utility>
#include <vector>
#include <ucontext.h>
#include <sys/mman.h>

using std::abs;
using std::ceil;
using std::isnan;
using std::sqrt;

// ---------------------------------------------------------------- decorations
#define __device__
#define __global__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __launch_bounds__(...)
#define __constant__

// ---------------------------------------------------------------- vector types
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int3 { int x, y, z; };
struct uint2 { unsigned int x, y; };
struct uint3 { unsigned int x, y, z; };
struct dim3 {
  unsigned int x, y, z;
  dim3(unsigned int x_ = 1, unsigned int y_ = 1, unsigned int z_ = 1) : x(x_), y(y_), z(z_) {}
};

// ---------------------------------------------------------------- math overloads of the CUDA headers (global namespace)
inline int min(int a, int b) { return a < b ? a : b; }
inline int max(int a, int b) { return a > b ? a : b; }
inline unsigned int min(unsigned int a, unsigned int b) { return a < b ? a : b; }
inline unsigned int max(unsigned int a, unsigned int b) { return a > b ? a : b; }
inline unsigned int min(int a, unsigned int b) { return min((unsigned int)a, b); }
inline unsigned int min(unsigned int a, int b) { return min(a, (unsigned int)b); }
inline unsigned int max(int a, unsigned int b) { return max((unsigned int)a, b); }
inline unsigned int max(unsigned int a, int b) { return max(a, (unsigned int)b); }
inline float min(float a, float b) { return fminf(a, b); }
inline float max(float a, float b) { return fmaxf(a, b); }
inline double min(double a, double b) { return fmin(a, b); }
inline double max(double a, double b) { return fmax(a, b); }
inline double min(float a, double b) { return fmin((double)a, b); }
inline double min(double a, float b) { return fmin(a, (double)b); }
inline double max(float a, double b) { return fmax((double)a, b); }
inline double max(double a, float b) { return fmax(a, (double)b); }

namespace cuda_on_host {

// ---------------------------------------------------------------- exp hook
typedef float (*exp_fn_t)(float);
inline exp_fn_t& exp_fn() { static exp_fn_t f = nullptr; return f; }
inline float exp_hook(float x) { exp_fn_t f = exp_fn(); return f ? f(x) : ::expf(x); }
inline double exp_hook(double x) { return ::exp(x); }

// ---------------------------------------------------------------- per-CUDA-thread context
struct ThreadCtx { dim3 blockIdx, threadIdx, blockDim, gridDim; };
inline ThreadCtx*& cur() { static thread_local ThreadCtx* c = nullptr; return c; }

struct BlockSched;
inline BlockSched*& sched() { static thread_local BlockSched* s = nullptr; return s; }

struct Fiber { ucontext_t ctx; bool finished; ThreadCtx tc; };

struct BlockSched {
  static constexpr size_t STACK = 512 * 1024;     // integrateCUDA keeps ~12 KB of per-thread arrays (forward.cu:1018,1156-1158,1242-1243)
  std::vector<Fiber> fibers;
  char* stacks = nullptr;
  size_t nstacks = 0;
  ucontext_t main_ctx;
  int running = -1;
  int count_cur = 0, count_prev = 0;
  const std::function<void()>* body = nullptr;

  ~BlockSched() { if (stacks) munmap(stacks, nstacks * STACK); }
  void reserve(size_t n) {
    if (n <= nstacks) return;
    if (stacks) munmap(stacks, nstacks * STACK);
    stacks = (char*)mmap(nullptr, n * STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (stacks == (char*)MAP_FAILED) { perror("cuda_on_host: mmap"); abort(); }
    nstacks = n;
    fibers.resize(n);
  }
  static void entry() {
    BlockSched* s = sched();
    (*s->body)();
    s->fibers[s->running].finished = true;
    swapcontext(&s->fibers[s->running].ctx, &s->main_ctx);
  }
  void yield() { swapcontext(&fibers[running].ctx, &main_ctx); }
  void run_block(const dim3& grid, const dim3& block, const dim3& bidx, const std::function<void()>& fn) {
    const size_t n = size_t(block.x) * block.y * block.z;
    reserve(n);
    body = &fn;
    size_t t = 0;
    for (unsigned z = 0; z < block.z; z++) for (unsigned y = 0; y < block.y; y++) for (unsigned x = 0; x < block.x; x++, t++) {
      Fiber& f = fibers[t];
      f.finished = false;
      f.tc.blockIdx = bidx; f.tc.threadIdx = dim3(x, y, z); f.tc.blockDim = block; f.tc.gridDim = grid;
      getcontext(&f.ctx);
      f.ctx.uc_stack.ss_sp = stacks + t * STACK;
      f.ctx.uc_stack.ss_size = STACK;
      f.ctx.uc_link = nullptr;
      makecontext(&f.ctx, (void (*)())entry, 0);
    }
    count_cur = count_prev = 0;
    size_t alive = n;
    while (alive) {                       // one pass = one barrier phase
      for (size_t i = 0; i < n; i++) {
        if (fibers[i].finished) continue;
        running = (int)i;
        cur() = &fibers[i].tc;
        swapcontext(&main_ctx, &fibers[i].ctx);
        if (fibers[i].finished) alive--;
      }
      count_prev = count_cur;
      count_cur = 0;
    }
    running = -1;
    cur() = nullptr;
  }
};

inline int& num_threads() { static int n = 1; return n; }

// Flat launches (1-D blocks: the per-Gaussian kernels, none of which synchronises) run their threads straight on the caller's
// stack; a barrier reached outside a fiber aborts, so the shortcut cannot silently change semantics.
inline void barrier_yield() {
  BlockSched* s = sched();
  if (!s || s->running < 0) { fprintf(stderr, "cuda_on_host: block barrier in a kernel launched with a 1-D block\n"); abort(); }
  s->yield();
}

inline void run_grid(dim3 grid, dim3 block, const std::function<void()>& fn) {
  const long nblocks = long(grid.x) * grid.y * grid.z;
  const bool flat = (block.y == 1 && block.z == 1);
  const int nt = std::max(1, num_threads());
#pragma omp parallel for schedule(dynamic, 1) num_threads(nt) if (nt > 1)
  for (long b = 0; b < nblocks; b++) {
    dim3 bidx((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / (long(grid.x) * grid.y)));
    if (flat) {
      ThreadCtx tc; tc.blockIdx = bidx; tc.blockDim = block; tc.gridDim = grid;
      ThreadCtx* saved = cur();
      cur() = &tc;
      for (unsigned x = 0; x < block.x; x++) { tc.threadIdx = dim3(x, 0, 0); fn(); }
      cur() = saved;
    } else {
      static thread_local BlockSched* s = nullptr;
      if (!s) s = new BlockSched();
      sched() = s;
      s->run_block(grid, block, bidx, fn);
    }
  }
}

template <class... A> struct Bound { dim3 g, b; std::tuple<A...> args; };
struct Cfg {
  dim3 g, b;
  template <class... A> Bound<A...> operator()(A... a) const { return Bound<A...>{g, b, std::tuple<A...>(a...)}; }
};
inline Cfg cfg(dim3 g, dim3 b) { return Cfg{g, b}; }

template <class... P, class... A, size_t... I>
inline void invoke(void (*k)(P...), const std::tuple<A...>& t, std::index_sequence<I...>) { k(std::get<I>(t)...); }

// Called after every kernel launch has run to completion (launches are synchronous here): lets the driver of the reference's code
// (ref_api.cpp) look at an array BETWEEN two kernels of one Rasterizer call -- the render kernel's raw opacity sums, which the
// next kernel rescales in place (backward.cu:395-403).
inline std::function<void()>& post_launch_hook() { static std::function<void()> h; return h; }

}  // namespace cuda_on_host

// kernel % cfg(grid, block)(args...)   ==   kernel<<<grid, block>>>(args...)
template <class... P, class... A>
inline void operator%(void (*k)(P...), const cuda_on_host::Bound<A...>& b) {
  static_assert(sizeof...(P) == sizeof...(A), "kernel launched with the wrong number of arguments");
  std::function<void()> fn = [&]() { cuda_on_host::invoke(k, b.args, std::index_sequence_for<A...>{}); };
  cuda_on_host::run_grid(b.g, b.b, fn);
  if (cuda_on_host::post_launch_hook()) cuda_on_host::post_launch_hook()();
}

// ---------------------------------------------------------------- cooperative groups
namespace cooperative_groups {
struct thread_block {
  dim3 group_index() const { return cuda_on_host::cur()->blockIdx; }
  dim3 thread_index() const { return cuda_on_host::cur()->threadIdx; }
  unsigned int thread_rank() const {
    const cuda_on_host::ThreadCtx* c = cuda_on_host::cur();
    return (c->threadIdx.z * c->blockDim.y + c->threadIdx.y) * c->blockDim.x + c->threadIdx.x;
  }
  void sync() const { cuda_on_host::barrier_yield(); }
};
struct grid_group {
  unsigned long long thread_rank() const {
    const cuda_on_host::ThreadCtx* c = cuda_on_host::cur();
    const unsigned long long bsz = (unsigned long long)c->blockDim.x * c->blockDim.y * c->blockDim.z;
    const unsigned long long brank = ((unsigned long long)c->blockIdx.z * c->gridDim.y + c->blockIdx.y) * c->gridDim.x + c->blockIdx.x;
    const unsigned long long trank = ((unsigned long long)c->threadIdx.z * c->blockDim.y + c->threadIdx.y) * c->blockDim.x + c->threadIdx.x;
    return brank * bsz + trank;
  }
};
inline thread_block this_thread_block() { return thread_block(); }
inline grid_group this_grid() { return grid_group(); }
}  // namespace cooperative_groups

inline int __syncthreads_count(int pred) {
  cuda_on_host::BlockSched* s = cuda_on_host::sched();
  if (!s || s->running < 0) { fprintf(stderr, "cuda_on_host: __syncthreads_count outside a fiber block\n"); abort(); }
  s->count_cur += pred ? 1 : 0;
  s->yield();
  return s->count_prev;
}
inline void __syncthreads() { cuda_on_host::barrier_yield(); }
inline void __trap() { fprintf(stderr, "cuda_on_host: __trap()\n"); abort(); }

inline float atomicAdd(float* addr, float v) {
  uint32_t* a = reinterpret_cast<uint32_t*>(addr);
  uint32_t old_bits = __atomic_load_n(a, __ATOMIC_RELAXED), new_bits;
  float old;
  do {
    memcpy(&old, &old_bits, 4);
    const float upd = old + v;
    memcpy(&new_bits, &upd, 4);
  } while (!__atomic_compare_exchange_n(a, &old_bits, new_bits, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
  return old;
}

// ---------------------------------------------------------------- runtime API used by rasterizer_impl.cu
typedef int cudaError_t;
enum { cudaSuccess = 0 };
enum cudaMemcpyKind { cudaMemcpyHostToHost, cudaMemcpyHostToDevice, cudaMemcpyDeviceToHost, cudaMemcpyDeviceToDevice };
inline cudaError_t cudaMemcpy(void* d, const void* s, size_t n, cudaMemcpyKind) { memcpy(d, s, n); return cudaSuccess; }
inline cudaError_t cudaMemset(void* d, int v, size_t n) { memset(d, v, n); return cudaSuccess; }
inline cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
inline const char* cudaGetErrorString(cudaError_t) { return "no error (cuda_on_host)"; }

// ---------------------------------------------------------------- CUB entry points (contract: see the header comment)
namespace cub {
struct DeviceScan {
  template <class In, class Out>
  static cudaError_t InclusiveSum(void* temp, size_t& temp_bytes, In in, Out out, int n) {
    if (!temp) { temp_bytes = 128; return cudaSuccess; }
    typename std::remove_reference<decltype(*out)>::type run = 0;
    for (int i = 0; i < n; i++) { run += in[i]; out[i] = run; }
    return cudaSuccess;
  }
};
struct DeviceRadixSort {
  template <class K, class V>
  static cudaError_t SortPairs(void* temp, size_t& temp_bytes, const K* kin, K* kout, const V* vin, V* vout, int n,
                               int begin_bit = 0, int end_bit = int(sizeof(K) * 8)) {
    if (!temp) { temp_bytes = 128; return cudaSuccess; }
    const int nb = end_bit - begin_bit;
    const K mask = nb >= int(sizeof(K) * 8) ? ~K(0) : ((K(1) << nb) - 1);
    std::vector<int> idx(n);
    std::iota(idx.begin(), idx.end(), 0);
    std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return ((kin[a] >> begin_bit) & mask) < ((kin[b] >> begin_bit) & mask); });
    for (int i = 0; i < n; i++) { kout[i] = kin[idx[i]]; vout[i] = vin[idx[i]]; }
    return cudaSuccess;
  }
};
}  // namespace cub

// `exp(x)` in the reference's kernels (forward.cu:565, backward.cu:852, forward.cu:1054,1322,1333) -> hook.  Function-like, defined
// after every standard header above has been included, so only the reference's own call sites see it.
#define exp(x) cuda_on_host::exp_hook(x)
