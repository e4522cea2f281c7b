// radegs_adam.hip -- multi-tensor Adam step: the update torch.optim.Adam(l, lr=0.0, eps=1e-15)
// (scene/gaussian_model.py:338-349) applies to the six per-Gaussian parameter groups every iteration, as ONE launch.
//   m += (1-b1) (g - m);  v = b2 v + (1-b2) g^2;  p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// 59 floats per Gaussian x (read p,g,m,v + write p,m,v) = 1.65 GB per step at P = 1M: pure HBM streaming.  Each block
// owns a 2048-element chunk of one tensor (block -> tensor by a prefix table in the kernel argument), 2 x dwordx4 per
// thread per array.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/radegs.h"

namespace rga {

constexpr int kMaxTensors = RADEGS_ADAM_MAX_TENSORS, kChunk = 2048;

struct Table {
  float* p[kMaxTensors]; const float* g[kMaxTensors]; float* m[kMaxTensors]; float* v[kMaxTensors];
  unsigned long long n[kMaxTensors];
  float step_size[kMaxTensors], bc2_sqrt[kMaxTensors];
  unsigned block_start[kMaxTensors + 1];
  int count;
  float omb1, beta2, omb2, eps;   // 1-beta1, beta2, 1-beta2 rounded from double like torch's Python scalars
};

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float omb1, float b2, float omb2, float eps, float step_size,
                                      float bc2_sqrt) {
  m = m + omb1 * (g - m);
  v = v * b2 + omb2 * (g * g);
  const float denom = sqrtf(v) / bc2_sqrt + eps;
  p = p - step_size * (m / denom);
}

__global__ void __launch_bounds__(256) adam_kernel(const Table t) {
  int ti = 0;
#pragma unroll 1
  while (ti + 1 < t.count && blockIdx.x >= t.block_start[ti + 1]) ti++;
  const unsigned long long base = (unsigned long long)(blockIdx.x - t.block_start[ti]) * kChunk;
  const unsigned long long n = t.n[ti];
  float* __restrict__ P = t.p[ti]; const float* __restrict__ G = t.g[ti]; float* __restrict__ M = t.m[ti]; float* __restrict__ V = t.v[ti];
  const float ss = t.step_size[ti], bc = t.bc2_sqrt[ti];
#pragma unroll
  for (int h = 0; h < 2; h++) {
    const unsigned long long i = base + (unsigned long long)h * 1024 + threadIdx.x * 4;
    if (i + 3 < n && ((((uintptr_t)P | (uintptr_t)G | (uintptr_t)M | (uintptr_t)V) & 15) == 0)) {
      float4 p = *reinterpret_cast<float4*>(P + i), m = *reinterpret_cast<float4*>(M + i), v = *reinterpret_cast<float4*>(V + i);
      const float4 g = *reinterpret_cast<const float4*>(G + i);
      adam1(p.x, g.x, m.x, v.x, t.omb1, t.beta2, t.omb2, t.eps, ss, bc); adam1(p.y, g.y, m.y, v.y, t.omb1, t.beta2, t.omb2, t.eps, ss, bc);
      adam1(p.z, g.z, m.z, v.z, t.omb1, t.beta2, t.omb2, t.eps, ss, bc); adam1(p.w, g.w, m.w, v.w, t.omb1, t.beta2, t.omb2, t.eps, ss, bc);
      *reinterpret_cast<float4*>(P + i) = p; *reinterpret_cast<float4*>(M + i) = m; *reinterpret_cast<float4*>(V + i) = v;
    } else {
      for (int k = 0; k < 4; k++)
        if (i + k < n) adam1(P[i + k], G[i + k], M[i + k], V[i + k], t.omb1, t.beta2, t.omb2, t.eps, ss, bc);
    }
  }
}

}  // namespace rga

extern "C" int radegs_adam_step(int count, const RadegsAdamTensor* tensors, double beta1, double beta2, double eps, void* stream) {
  if (count < 0 || count > rga::kMaxTensors || (count > 0 && !tensors)) return RADEGS_ERR_INVALID_ARG;
  rga::Table t{};
  unsigned blocks = 0;
  int k = 0;
  for (int i = 0; i < count; i++) {
    const RadegsAdamTensor& a = tensors[i];
    if (a.numel == 0) continue;
    if (!a.param || !a.grad || !a.exp_avg || !a.exp_avg_sq || a.step < 1.0) return RADEGS_ERR_INVALID_ARG;
    t.p[k] = a.param; t.g[k] = a.grad; t.m[k] = a.exp_avg; t.v[k] = a.exp_avg_sq; t.n[k] = a.numel;
    // bias corrections in double on the host, like torch's Python scalars (torch/optim/adam.py, _single_tensor_adam)
    const double bc1 = 1.0 - pow(beta1, a.step), bc2 = 1.0 - pow(beta2, a.step);
    t.step_size[k] = (float)((double)a.lr / bc1);
    t.bc2_sqrt[k] = (float)sqrt(bc2);
    t.block_start[k] = blocks;
    blocks += (unsigned)((a.numel + rga::kChunk - 1) / rga::kChunk);
    k++;
  }
  t.block_start[k] = blocks;
  t.count = k; t.omb1 = (float)(1.0 - beta1); t.beta2 = (float)beta2; t.omb2 = (float)(1.0 - beta2); t.eps = (float)eps;
  if (k == 0) return 0;
  hipLaunchKernelGGL(rga::adam_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), t);
  return hipGetLastError() == hipSuccess ? 0 : RADEGS_ERR_HIP;
}
