// radegs_photometric.hip -- the photometric loss that closes every training iteration:
//     rgb_loss = (1 - lambda) * l1_loss(image, gt) + lambda * (1 - ssim(image, gt))             train.py:159
//     l1_loss / ssim (11x11 Gaussian window, sigma 1.5, zero padding, per channel)              utils/loss_utils.py:17-63
// forward and the gradient w.r.t. `image`.  The reference runs 5 grouped conv2d + ~15 elementwise kernels forward and
// the autograd mirror of all of them backward.  Here: one tiled kernel per direction.
//
// Tiling for gfx950: a 256-thread block owns a 16x64 output tile of one channel.  The 26x74 input tile (halo 5) of both
// images is staged in LDS once; the separable blur runs horizontally for the 5 moment maps (a, b, a^2, b^2, ab) into LDS
// and vertically into registers (lanes walk columns: stride-1 LDS reads, no bank conflicts).  The forward also stores the
// three per-pixel derivative maps d ssim/d(mu1, E[a^2], E[ab]); the backward blurs those with the same (symmetric)
// window and combines them with the pixel's own values:   d ssim/d a(q) = B[D_mu](q) + 2 a(q) B[D_11](q) + b(q) B[D_12](q).
// Sums (L1, SSIM) are reduced wave -> block -> fixed-order final pass in double: deterministic, no atomics.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "../../include/radegs.h"

namespace rgp {

constexpr int R = 5, TY = 16, TX = 64, IY = TY + 2 * R, IX = TX + 2 * R, IXP = IX + 2;
struct Win { float w[11]; };

struct Img { int W, H, C; };

__device__ __forceinline__ void stage_tile(const float* __restrict__ src, const Img& im, int ch, int ty0, int tx0, float (*dst)[IXP], int tid) {
  const size_t base = (size_t)ch * im.H * im.W;
  for (int idx = tid; idx < IY * IX; idx += 256) {
    const int r = idx / IX, c = idx - r * IX;
    const int y = ty0 + r - R, x = tx0 + c - R;
    dst[r][c] = (y >= 0 && y < im.H && x >= 0 && x < im.W) ? src[base + (size_t)y * im.W + x] : 0.0f;
  }
}

__global__ void __launch_bounds__(256) photometric_fwd_kernel(const Img im, const Win win, const float* __restrict__ img, const float* __restrict__ gt,
                                                             float* __restrict__ dmaps /* [3,C,H,W] or null */, double* __restrict__ partial) {
  __shared__ float sa[IY][IXP], sb[IY][IXP];
  __shared__ float h[5][IY][TX];
  __shared__ float red[4][2];
  const int tid = threadIdx.x, ch = blockIdx.z;
  const int ty0 = blockIdx.y * TY, tx0 = blockIdx.x * TX;
  stage_tile(img, im, ch, ty0, tx0, sa, tid);
  stage_tile(gt, im, ch, ty0, tx0, sb, tid);
  __syncthreads();
  for (int idx = tid; idx < IY * TX; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    float m1 = 0.f, m2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {
      const float a = sa[r][c + k], b = sb[r][c + k], w = win.w[k];
      m1 += w * a; m2 += w * b; e11 += w * (a * a); e22 += w * (b * b); e12 += w * (a * b);
    }
    h[0][r][c] = m1; h[1][r][c] = m2; h[2][r][c] = e11; h[3][r][c] = e22; h[4][r][c] = e12;
  }
  __syncthreads();
  const int c = tid & 63, r0 = tid >> 6;
  const size_t HW = (size_t)im.H * im.W, CHW = HW * im.C;
  float sum_l1 = 0.f, sum_ssim = 0.f;
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const int r = r0 + 4 * s;
    const int y = ty0 + r, x = tx0 + c;
    if (y >= im.H || x >= im.W) continue;
    float mu1 = 0.f, mu2 = 0.f, e11 = 0.f, e22 = 0.f, e12 = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {
      const float w = win.w[k];
      mu1 += w * h[0][r + k][c]; mu2 += w * h[1][r + k][c]; e11 += w * h[2][r + k][c]; e22 += w * h[3][r + k][c]; e12 += w * h[4][r + k][c];
    }
    const float C1 = 0.01f * 0.01f, C2 = 0.03f * 0.03f;
    const float s11 = e11 - mu1 * mu1, s22 = e22 - mu2 * mu2, s12 = e12 - mu1 * mu2;
    const float A1 = 2.f * mu1 * mu2 + C1, A2 = 2.f * s12 + C2, B1 = mu1 * mu1 + mu2 * mu2 + C1, B2 = s11 + s22 + C2;
    const float inv = 1.0f / (B1 * B2);
    sum_ssim += (A1 * A2) * inv;
    sum_l1 += fabsf(sa[r + R][c + R] - sb[r + R][c + R]);
    if (dmaps) {
      const size_t i = (size_t)ch * HW + (size_t)y * im.W + x;
      dmaps[i] = ((2.f * mu2 * (A2 - A1)) * (B1 * B2) - (A1 * A2) * (2.f * mu1 * (B2 - B1))) * inv * inv;
      dmaps[CHW + i] = -(A1 * A2) * inv / B2;
      dmaps[2 * CHW + i] = 2.f * A1 * inv;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { sum_l1 += __shfl_xor(sum_l1, o); sum_ssim += __shfl_xor(sum_ssim, o); }
  if ((tid & 63) == 0) { red[tid >> 6][0] = sum_l1; red[tid >> 6][1] = sum_ssim; }
  __syncthreads();
  if (tid == 0) {
    const size_t b = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
    partial[2 * b] = (double)red[0][0] + (double)red[1][0] + (double)red[2][0] + (double)red[3][0];
    partial[2 * b + 1] = (double)red[0][1] + (double)red[1][1] + (double)red[2][1] + (double)red[3][1];
  }
}

__global__ void __launch_bounds__(256) photometric_final_kernel(const double* __restrict__ partial, int nblocks, double inv_n, float lambda,
                                                               float* __restrict__ out3 /* loss, l1, ssim */) {
  __shared__ double s0[256], s1[256];
  double a = 0.0, b = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) { a += partial[2 * (size_t)i]; b += partial[2 * (size_t)i + 1]; }
  s0[threadIdx.x] = a; s1[threadIdx.x] = b;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) { s0[threadIdx.x] += s0[threadIdx.x + o]; s1[threadIdx.x] += s1[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float l1 = (float)(s0[0] * inv_n), ss = (float)(s1[0] * inv_n);
    out3[0] = (1.0f - lambda) * l1 + lambda * (1.0f - ss);
    out3[1] = l1; out3[2] = ss;
  }
}

// grad_img = coef[0] * sign(a-b)/n + coef[1] * d(mean ssim)/d a     (coef: device floats)
__global__ void __launch_bounds__(256) photometric_bwd_kernel(const Img im, const Win win, const float* __restrict__ img, const float* __restrict__ gt,
                                                             const float* __restrict__ dmaps, const float* __restrict__ coef, float inv_n,
                                                             float* __restrict__ grad) {
  __shared__ float sd[3][IY][IXP];
  __shared__ float h[3][IY][TX];
  const int tid = threadIdx.x, ch = blockIdx.z;
  const int ty0 = blockIdx.y * TY, tx0 = blockIdx.x * TX;
  const size_t HW = (size_t)im.H * im.W, CHW = HW * im.C;
#pragma unroll
  for (int q = 0; q < 3; q++) stage_tile(dmaps + q * CHW, im, ch, ty0, tx0, sd[q], tid);
  __syncthreads();
  for (int idx = tid; idx < IY * TX; idx += 256) {
    const int r = idx >> 6, c = idx & 63;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {
      const float w = win.w[k];
      v0 += w * sd[0][r][c + k]; v1 += w * sd[1][r][c + k]; v2 += w * sd[2][r][c + k];
    }
    h[0][r][c] = v0; h[1][r][c] = v1; h[2][r][c] = v2;
  }
  __syncthreads();
  const int c = tid & 63, r0 = tid >> 6;
  const float c_l1 = coef[0] * inv_n, c_ss = coef[1] * inv_n;
#pragma unroll
  for (int s = 0; s < 4; s++) {
    const int r = r0 + 4 * s;
    const int y = ty0 + r, x = tx0 + c;
    if (y >= im.H || x >= im.W) continue;
    float v0 = 0.f, v1 = 0.f, v2 = 0.f;
#pragma unroll
    for (int k = 0; k < 11; k++) {
      const float w = win.w[k];
      v0 += w * h[0][r + k][c]; v1 += w * h[1][r + k][c]; v2 += w * h[2][r + k][c];
    }
    const size_t i = (size_t)ch * HW + (size_t)y * im.W + x;
    const float a = img[i], b = gt[i];
    const float d = a - b;
    const float sgn = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
    grad[i] = c_l1 * sgn + c_ss * (v0 + 2.f * a * v1 + b * v2);
  }
}

static Win make_window() {
  // loss_utils.py:23-25: float32 tensor of exp(-(x-5)^2 / (2*1.5^2)), divided by its float32 sum
  Win w;
  float sum = 0.f;
  for (int x = 0; x < 11; x++) { w.w[x] = (float)exp(-(double)((x - 5) * (x - 5)) / (2.0 * 1.5 * 1.5)); }
  for (int x = 0; x < 11; x++) sum += w.w[x];
  for (int x = 0; x < 11; x++) w.w[x] /= sum;
  return w;
}

}  // namespace rgp

using namespace rgp;

extern "C" {

size_t radegs_photometric_scratch_bytes(int width, int height, int channels) {
  return (size_t)((width + TX - 1) / TX) * ((height + TY - 1) / TY) * channels * 2 * sizeof(double);
}

int radegs_photometric_forward(int width, int height, int channels, const float* image, const float* gt, float lambda_dssim, void* scratch,
                               float* dmaps, float* out_loss3, void* stream_v) {
  if (width <= 0 || height <= 0 || channels <= 0 || !image || !gt || !scratch || !out_loss3) return RADEGS_ERR_INVALID_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream_v);
  const Img im{width, height, channels};
  const dim3 grid((width + TX - 1) / TX, (height + TY - 1) / TY, channels);
  hipLaunchKernelGGL(photometric_fwd_kernel, grid, dim3(256), 0, s, im, make_window(), image, gt, dmaps, static_cast<double*>(scratch));
  hipLaunchKernelGGL(photometric_final_kernel, dim3(1), dim3(256), 0, s, static_cast<const double*>(scratch), (int)(grid.x * grid.y * grid.z),
                     1.0 / ((double)width * height * channels), lambda_dssim, out_loss3);
  return hipGetLastError() == hipSuccess ? 0 : RADEGS_ERR_HIP;
}

int radegs_photometric_backward(int width, int height, int channels, const float* image, const float* gt, const float* dmaps,
                                const float* coef2, float* grad_image, void* stream_v) {
  if (width <= 0 || height <= 0 || channels <= 0 || !image || !gt || !dmaps || !coef2 || !grad_image) return RADEGS_ERR_INVALID_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream_v);
  const Img im{width, height, channels};
  const dim3 grid((width + TX - 1) / TX, (height + TY - 1) / TY, channels);
  hipLaunchKernelGGL(photometric_bwd_kernel, grid, dim3(256), 0, s, im, make_window(), image, gt, dmaps, coef2,
                     (float)(1.0 / ((double)width * height * channels)), grad_image);
  return hipGetLastError() == hipSuccess ? 0 : RADEGS_ERR_HIP;
}

}  // extern "C"
