// TEST INFRASTRUCTURE: tiny CUDA-style kernels (this repository's own code, written the way the reference writes its kernels) that
// exercise the host execution model of oracle/ref_shim/cuda_on_host.h by itself: __shared__ staging + block.sync(), __syncthreads_count,
// divergent early exits after the last barrier, float atomics, a flat 1-D launch, the CUDA min/max overloads and the CUB stand-ins.
// Built by tests/test_ref_shim.py with the same launch-syntax substitution oracle/build_ref.py applies to the reference's sources.
#include <cuda_runtime.h>
#include <cooperative_groups.h>
#include <cub/cub.cuh>
namespace cg = cooperative_groups;

// out[b] = sum of in over block b's 256 elements, computed through shared memory in two barrier phases; votes[b] = how many
// threads of the block saw a positive element (via __syncthreads_count)
__global__ void block_sum_kernel(const float* in, float* out, int* votes, int n) {
  auto block = cg::this_thread_block();
  __shared__ float stage[256];
  const unsigned t = block.thread_rank();
  const unsigned i = (block.group_index().y * 3 + block.group_index().x) * 256 + t;   // launched on a (3, gy) grid of (16,16) blocks
  stage[t] = i < (unsigned)n ? in[i] : 0.f;
  const int positive = __syncthreads_count(i < (unsigned)n && in[i] > 0.f);
  if (t >= 16) return;                       // most threads leave; the rest must still see every staged value
  float s = 0.f;
  for (int k = 0; k < 16; k++) s += stage[16 * k + t];
  stage[t] = s;                              // slot t (< 16) was read by thread t alone: no hazard; the barrier below orders the column sums
  block.sync();                              // (the 240 threads that returned count as arrived, as on the device)
  if (t == 0) {
    float tot = 0.f;
    for (int k = 0; k < 16; k++) tot += stage[k];
    out[block.group_index().y * 3 + block.group_index().x] = tot;
    votes[block.group_index().y * 3 + block.group_index().x] = positive;
  }
}

// every thread adds its value to one of 7 bins (float atomics), flat launch
__global__ void scatter_add_kernel(const float* in, float* bins, int n) {
  auto idx = cg::this_grid().thread_rank();
  if (idx >= n) return;
  atomicAdd(&bins[idx % 7], in[idx]);
}

extern "C" {
void shim_block_sum(const float* in, float* out, int* votes, int n, int gy) {
  dim3 grid(3, gy, 1), block(16, 16, 1);
  block_sum_kernel<<<grid, block>>>(in, out, votes, n);
}
void shim_scatter_add(const float* in, float* bins, int n) { scatter_add_kernel << <(n + 255) / 256, 256 >> > (in, bins, n); }
// CUDA overload resolution the reference relies on (auxiliary.h:62-72, forward.cu:119-120)
void shim_minmax(unsigned* u, double* d) {
  unsigned gridx = 10;
  u[0] = min(gridx, max((int)0, (int)(-3.5f)));          // (unsigned, int) -> unsigned: 0
  u[1] = min(gridx, max((int)0, (int)(123.7f)));         // 10
  float f = 3.0e-7f;
  d[0] = max(1e-6, f);                                   // (double, float) -> double: 1e-6 exactly
  d[1] = (double)(float)max(1e-6, 2.5f * f);             // still the double literal, rounded once to float
}
void shim_sort_pairs(const unsigned long long* kin, unsigned long long* kout, const unsigned* vin, unsigned* vout, int n, int end_bit) {
  size_t bytes = 0;
  cub::DeviceRadixSort::SortPairs(nullptr, bytes, kin, kout, vin, vout, n);
  char tmp[256];
  cub::DeviceRadixSort::SortPairs(tmp, bytes, kin, kout, vin, vout, n, 0, end_bit);
}
void shim_inclusive_sum(const unsigned* in, unsigned* out, int n) {
  size_t bytes = 0;
  cub::DeviceScan::InclusiveSum(nullptr, bytes, in, out, n);
  char tmp[256];
  cub::DeviceScan::InclusiveSum(tmp, bytes, in, out, n);
}
void shim_set_threads(int n) { cuda_on_host::num_threads() = n; }
}
