// Rate and placement of fp32 atomic line updates by memory scope on gfx950 (measurement tool, not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o /tmp/atomic_scope scripts/ubench/atomic_scope.hip && /tmp/atomic_scope
// The access pattern is the blend backward's: every 16-lane DPP row of a wave adds 16 consecutive floats (one 64-B line) to a
// pseudo-random accumulator line.  Questions: (1) how many line updates per second at agent scope (memory side on this multi-XCD
// part) against workgroup scope (the issuing XCD's L2); (2) are workgroup-scope updates exact when every XCD owns a private copy
// chosen by the hardware XCC id; (3) does blockIdx % 8 name the XCD a workgroup runs on.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t xcc_id() {
  uint32_t v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

// SCOPE: 0 agent, 1 workgroup, 2 wavefront.  copies: 1 (shared) or 8 (private per XCC id).  window: lines a workgroup scatters over
// (0 = the whole copy).
template <int SCOPE>
__global__ void __launch_bounds__(256) k_atomic(float* acc, uint32_t lines, int iters, int copies, uint32_t window, uint32_t* xcc_hist) {
  const uint32_t lane = threadIdx.x & 63, row = lane >> 4, l16 = lane & 15;
  const uint32_t wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const uint32_t xcc = xcc_id();
  if (threadIdx.x == 0 && xcc_hist) atomicAdd(&xcc_hist[(blockIdx.x & 7) * 16 + xcc], 1u);
  float* base = acc + (size_t)(copies > 1 ? xcc : 0) * lines * 16;
  const uint32_t w0 = window ? (mix(blockIdx.x) % (lines - window)) : 0, span = window ? window : lines;
  for (int i = 0; i < iters; i++) {
    uint32_t h = w0 + mix((wave * 4 + row) * 7919u + i) % span;
    float* p = base + (size_t)h * 16 + l16;
    if (SCOPE == 0) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (SCOPE == 1) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (SCOPE == 2) __hip_atomic_fetch_add(p, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
  }
}

__global__ void k_sum(const float* acc, size_t n, double* out) {
  double s = 0;
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += acc[i];
  atomicAdd(out, s);
}

int main() {
  const uint32_t lines = 1u << 20;   // 64 MB per copy, as for 1 M Gaussians
  const int grid = 4096, iters = 128;
  const double updates = (double)grid * 4 * 4 * iters;   // line updates
  float* acc; double* dsum; uint32_t* hist;
  CK(hipMalloc(&acc, (size_t)8 * lines * 64));
  CK(hipMalloc(&dsum, 8));
  CK(hipMalloc(&hist, 8 * 16 * 4));
  CK(hipMemset(hist, 0, 8 * 16 * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const char* names[3] = {"agent", "workgroup", "wavefront"};
  for (int copies : {1, 8})
    for (uint32_t window : {0u, 4096u})
      for (int scope = 0; scope < 3; scope++) {
        float best = 1e9f; double sum = 0;
        for (int rep = 0; rep < 3; rep++) {
          CK(hipMemset(acc, 0, (size_t)8 * lines * 64));
          CK(hipMemset(dsum, 0, 8));
          CK(hipDeviceSynchronize());
          CK(hipEventRecord(e0));
          uint32_t* hp = (rep == 0 && scope == 0 && copies == 1 && window == 0) ? hist : nullptr;
          if (scope == 0) k_atomic<0><<<grid, 256>>>(acc, lines, iters, copies, window, hp);
          if (scope == 1) k_atomic<1><<<grid, 256>>>(acc, lines, iters, copies, window, hp);
          if (scope == 2) k_atomic<2><<<grid, 256>>>(acc, lines, iters, copies, window, hp);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
          float ms; CK(hipEventElapsedTime(&ms, e0, e1));
          if (ms < best) best = ms;
          k_sum<<<1024, 256>>>(acc, (size_t)8 * lines * 16, dsum);
          CK(hipMemcpy(&sum, dsum, 8, hipMemcpyDeviceToHost));
        }
        printf("copies %d window %5u scope %-9s : %.3f ms  %.2f G line updates/s   sum/expected %.6f\n", copies, window, names[scope], best,
               updates / best * 1e-6, sum / (updates * 16));
      }
  std::vector<uint32_t> h(128);
  CK(hipMemcpy(h.data(), hist, 512, hipMemcpyDeviceToHost));
  printf("workgroups by (blockIdx %% 8) x XCC id:\n");
  for (int b = 0; b < 8; b++) { for (int x = 0; x < 16; x++) printf("%5u", h[b * 16 + x]); printf("\n"); }
  return 0;
}
