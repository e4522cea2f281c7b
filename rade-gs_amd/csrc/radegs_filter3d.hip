// radegs_filter3d.hip -- the step that PRECEDES the rasterizer in every render() call: activations of the
// raw scaling / opacity parameters fused with the 3D (mip) filter,
//     GaussianModel.get_scaling_n_opacity_with_3D_filter    scene/gaussian_model.py:156-166
//       s = exp(_scaling);  det1 = prod s^2;  s'^2 = s^2 + filter_3D^2;  det2 = prod s'^2
//       scales = sqrt(s'^2);  opacity = sigmoid(_opacity) * sqrt(det1/det2)
// forward and backward.  The reference runs 10 elementwise torch kernels forward and ~20 in autograd backward over P
// Gaussians; both directions are one streaming kernel here (36 B / 52 B per Gaussian): HBM-bound by construction.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/radegs.h"

namespace rgf {

__device__ __forceinline__ float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

__global__ void __launch_bounds__(256) filter3d_fwd_kernel(int P, const float* __restrict__ scaling_raw, const float* __restrict__ opacity_raw,
                                                          const float* __restrict__ filter_3D, float* __restrict__ scales_out,
                                                          float* __restrict__ opacity_out) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const float f = filter_3D[i], f2 = f * f;
  float det1 = 1.0f, det2 = 1.0f;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float s = expf(scaling_raw[3 * (size_t)i + k]);
    const float s2 = s * s, a2 = s2 + f2;
    det1 *= s2; det2 *= a2;
    scales_out[3 * (size_t)i + k] = sqrtf(a2);
  }
  opacity_out[i] = sigmoidf(opacity_raw[i]) * sqrtf(det1 / det2);
}

__global__ void __launch_bounds__(256) filter3d_bwd_kernel(int P, const float* __restrict__ scaling_raw, const float* __restrict__ opacity_raw,
                                                          const float* __restrict__ filter_3D, const float* __restrict__ g_scales,
                                                          const float* __restrict__ g_opacity, float* __restrict__ g_scaling_raw,
                                                          float* __restrict__ g_opacity_raw) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  const float f = filter_3D[i], f2 = f * f;
  float s2[3], a2[3], det1 = 1.0f, det2 = 1.0f;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float s = expf(scaling_raw[3 * (size_t)i + k]);
    s2[k] = s * s; a2[k] = s2[k] + f2;
    det1 *= s2[k]; det2 *= a2[k];
  }
  const float coef = sqrtf(det1 / det2);
  const float sg = sigmoidf(opacity_raw[i]);
  const float go = g_opacity ? g_opacity[i] : 0.0f;
  g_opacity_raw[i] = go * coef * sg * (1.0f - sg);
  // d coef / d raw_k = coef * f^2 / (s_k^2 + f^2);  d scales_k / d raw_k = s_k^2 / sqrt(s_k^2 + f^2)
  const float gc = go * sg * coef;
#pragma unroll
  for (int k = 0; k < 3; k++) {
    const float gs = g_scales ? g_scales[3 * (size_t)i + k] : 0.0f;
    g_scaling_raw[3 * (size_t)i + k] = gs * s2[k] / sqrtf(a2[k]) + gc * f2 / a2[k];
  }
}

}  // namespace rgf

extern "C" {

int radegs_filter3d_forward(int P, const float* scaling_raw, const float* opacity_raw, const float* filter_3D, float* scales_out,
                            float* opacity_out, void* stream) {
  if (P < 0) return RADEGS_ERR_INVALID_ARG;
  if (P == 0) return 0;
  if (!scaling_raw || !opacity_raw || !filter_3D || !scales_out || !opacity_out) return RADEGS_ERR_INVALID_ARG;
  hipLaunchKernelGGL(rgf::filter3d_fwd_kernel, dim3((P + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), P, scaling_raw,
                     opacity_raw, filter_3D, scales_out, opacity_out);
  return hipGetLastError() == hipSuccess ? 0 : RADEGS_ERR_HIP;
}

int radegs_filter3d_backward(int P, const float* scaling_raw, const float* opacity_raw, const float* filter_3D, const float* grad_scales,
                             const float* grad_opacity, float* grad_scaling_raw, float* grad_opacity_raw, void* stream) {
  if (P < 0) return RADEGS_ERR_INVALID_ARG;
  if (P == 0) return 0;
  if (!scaling_raw || !opacity_raw || !filter_3D || !grad_scaling_raw || !grad_opacity_raw) return RADEGS_ERR_INVALID_ARG;
  hipLaunchKernelGGL(rgf::filter3d_bwd_kernel, dim3((P + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), P, scaling_raw,
                     opacity_raw, filter_3D, grad_scales, grad_opacity, grad_scaling_raw, grad_opacity_raw);
  return hipGetLastError() == hipSuccess ? 0 : RADEGS_ERR_HIP;
}

}  // extern "C"
