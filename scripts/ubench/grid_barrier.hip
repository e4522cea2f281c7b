// What does a grid-wide barrier cost on gfx950 (8 XCDs, device-scope atomics resolve at the memory side)?  Measurement tool, not part
// of the product: it decides whether a persistent radix-sort kernel with barriers between histogram / scan / scatter can beat the
// ~5 us a dependent kernel boundary costs.
//   hipcc --offload-arch=gfx950 -O3 -o scripts/ubench/grid_barrier scripts/ubench/grid_barrier.hip && scripts/ubench/grid_barrier
// Variants: flat (every workgroup adds to ONE counter and polls it), tree (one counter per XCD = blockIdx % 8, the last arriver of an
// XCD adds to the root, everybody polls a generation word), and for comparison a chain of empty dependent kernels.
// Every spin is bounded: a barrier that does not complete sets `fail` and the kernel leaves.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr uint32_t kSpinMax = 1u << 22;

__device__ __forceinline__ uint32_t ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT); }

// flat: ctr counts arrivals forever; barrier k is complete when ctr >= k * nblocks
__global__ void __launch_bounds__(256) k_flat(uint32_t* ctr, int nbar, uint32_t* fail, uint32_t* sink) {
  uint32_t acc = 0;
  for (int k = 1; k <= nbar; k++) {
    __syncthreads();
    if (threadIdx.x == 0) {
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const uint32_t target = (uint32_t)k * gridDim.x;
      uint32_t spins = 0;
      while (ld(ctr) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > kSpinMax) { *fail = 1u; break; }
      }
    }
    __syncthreads();
    acc += k;
  }
  if (threadIdx.x == 0 && acc == 0xFFFFFFFFu) *sink = acc;
}

// tree: xcd[8 * 16] per-XCD arrival counters (one 64-B line each), root counter, generation word
__global__ void __launch_bounds__(256) k_tree(uint32_t* xcd, uint32_t* root, uint32_t* gen, int nbar, uint32_t* fail, uint32_t* sink) {
  const uint32_t x = blockIdx.x & 7u;
  const uint32_t per_xcd = (gridDim.x + 7u - x) / 8u;   // workgroups with blockIdx % 8 == x
  uint32_t acc = 0;
  for (int k = 1; k <= nbar; k++) {
    __syncthreads();
    if (threadIdx.x == 0) {
      const uint32_t a = __hip_atomic_fetch_add(xcd + 16 * x, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
      if (a + 1u == (uint32_t)k * per_xcd) {   // last of this XCD
        const uint32_t r = __hip_atomic_fetch_add(root, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (r + 1u == (uint32_t)k * 8u) __hip_atomic_store(gen, (uint32_t)k, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      }
      uint32_t spins = 0;
      while (ld(gen) < (uint32_t)k) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > kSpinMax) { *fail = 1u; break; }
      }
    }
    __syncthreads();
    acc += k;
  }
  if (threadIdx.x == 0 && acc == 0xFFFFFFFFu) *sink = acc;
}

__global__ void __launch_bounds__(256) k_empty(uint32_t* sink) { if (threadIdx.x == 999) *sink = 1; }

int main() {
  uint32_t* d;
  CK(hipMalloc(&d, 4096));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int nbar = 200;
  for (int grid : {128, 256, 512, 1024}) {
    for (int variant = 0; variant < 2; variant++) {
      float best = 1e9f;
      uint32_t fail = 0;
      for (int rep = 0; rep < 5; rep++) {
        CK(hipMemset(d, 0, 4096));
        CK(hipEventRecord(e0));
        if (variant == 0) hipLaunchKernelGGL(k_flat, dim3(grid), dim3(256), 0, 0, d, nbar, d + 512, d + 513);
        else hipLaunchKernelGGL(k_tree, dim3(grid), dim3(256), 0, 0, d, d + 256, d + 272, nbar, d + 512, d + 513);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
        CK(hipMemcpy(&fail, d + 512, 4, hipMemcpyDeviceToHost));
      }
      printf("grid %4d  %s barrier: %.2f us each (%d barriers, best of 5)%s\n", grid, variant ? "tree" : "flat", 1000.f * best / nbar, nbar,
             fail ? "  [a spin ran out!]" : "");
    }
  }
  {
    float best = 1e9f;
    for (int rep = 0; rep < 5; rep++) {
      CK(hipEventRecord(e0));
      for (int i = 0; i < nbar; i++) hipLaunchKernelGGL(k_empty, dim3(512), dim3(256), 0, 0, d + 513);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    printf("chain of %d empty dependent kernels (512 x 256 threads): %.2f us each\n", nbar, 1000.f * best / nbar);
  }
  return 0;
}
