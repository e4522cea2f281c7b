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

// ---- GaussianModel.compute_3D_filter (scene/gaussian_model.py:179-232): per Gaussian, the smallest view-space depth over
// all training cameras that see it (z > 0.2 and inside the image enlarged by 15 %), turned into a filter radius.  The
// reference loops over cameras in Python (~15 torch kernels per camera, every 100 iterations); here every Gaussian walks
// the camera table (16 floats per camera, wave-uniform scalar loads) in one kernel.
struct CamRow { float r[9]; float t[3]; float fx, fy, w, h; };   // R row-major as stored (p_cam = p @ R + T)

__global__ void __launch_bounds__(256) filter3d_distance_kernel(int P, const float* __restrict__ xyz, int ncam, const CamRow* __restrict__ cams,
                                                               float* __restrict__ distance, unsigned* __restrict__ max_valid_bits) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  float best = 100000.0f;
  bool any = false;
  if (i < P) {
    const float x = xyz[3 * (size_t)i], y = xyz[3 * (size_t)i + 1], z = xyz[3 * (size_t)i + 2];
    for (int c = 0; c < ncam; c++) {
      const CamRow cam = cams[c];
      const float cx = x * cam.r[0] + y * cam.r[3] + z * cam.r[6] + cam.t[0];
      const float cy = x * cam.r[1] + y * cam.r[4] + z * cam.r[7] + cam.t[1];
      const float cz0 = x * cam.r[2] + y * cam.r[5] + z * cam.r[8] + cam.t[2];
      const bool valid_depth = cz0 > 0.2f;
      const float cz = fmaxf(cz0, 0.001f);
      const float px = cx / cz * cam.fx + cam.w / 2.0f, py = cy / cz * cam.fy + cam.h / 2.0f;
      const bool in_screen = px >= -0.15f * cam.w && px <= cam.w * 1.15f && py >= -0.15f * cam.h && py <= 1.15f * cam.h;
      if (valid_depth && in_screen) { best = fminf(best, cz); any = true; }
    }
    distance[i] = any ? best : -1.0f;   // -1 marks "seen by no camera" for the second pass
  }
  // max over the valid distances (positive floats order like their bit patterns)
  unsigned bits = (i < P && any) ? __float_as_uint(best) : 0u;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) bits = max(bits, (unsigned)__shfl_xor((int)bits, o));
  if ((threadIdx.x & 63) == 0 && bits) atomicMax(max_valid_bits, bits);
}

__global__ void __launch_bounds__(256) filter3d_finish_kernel(int P, const float* __restrict__ distance, const unsigned* __restrict__ max_valid_bits,
                                                             float focal_length, float* __restrict__ filter_3D) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  float d = distance[i];
  if (d < 0.0f) d = __uint_as_float(*max_valid_bits);
  filter_3D[i] = d / focal_length * 0.44721359549995793f;  // (0.2 ** 0.5)
}

}  // namespace rgf

extern "C" {

int radegs_compute_filter3d(int P, const float* xyz, int ncam, const float* cameras16, float focal_length, float* scratch_distance,
                            unsigned* scratch_max, float* filter_3D, void* stream_v) {
  if (P < 0 || ncam < 0) return RADEGS_ERR_INVALID_ARG;
  if (P == 0) return 0;
  if (!xyz || (ncam > 0 && !cameras16) || !scratch_distance || !scratch_max || !filter_3D || !(focal_length > 0.0f)) return RADEGS_ERR_INVALID_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream_v);
  if (hipMemsetAsync(scratch_max, 0, sizeof(unsigned), s) != hipSuccess) return RADEGS_ERR_HIP;
  hipLaunchKernelGGL(rgf::filter3d_distance_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, xyz, ncam,
                     reinterpret_cast<const rgf::CamRow*>(cameras16), scratch_distance, scratch_max);
  hipLaunchKernelGGL(rgf::filter3d_finish_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, scratch_distance, scratch_max, focal_length, filter_3D);
  return hipGetLastError() == hipSuccess ? 0 : RADEGS_ERR_HIP;
}

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
