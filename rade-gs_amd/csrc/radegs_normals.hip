// radegs_normals.hip -- the step that follows the rasterizer in every regularised training iteration:
// two depth (or coordinate) maps -> two normal maps by central differences, and the normal-consistency loss against the
// rendered normal map, forward and backward.
//
//   depths_double_to_points   utils/graphics_utils.py:97-112     P_k(y,x) = depth_k(y,x) * K^-1 [x+.5, y+.5, 1]
//   point_double_to_normal    utils/graphics_utils.py:116-123    N_k = normalize((P(y+1,x)-P(y-1,x)) x (P(y,x+1)-P(y,x-1))), border 0
//   depth_double_to_normal    utils/graphics_utils.py:125-127
//   loss                      train.py:152-155                   (1-r) mean(1 - n.N_0) + r mean(1 - n.N_1)
//
// The reference runs ~10 full-image torch kernels forward and ~25 backward (autograd) on this step.  Everything here is a
// 5-point stencil over a few float maps: pure HBM streaming, one kernel each way.  Thread per pixel, 64x4 blocks (rows of
// 64 consecutive pixels per wave => coalesced 256-B loads; vertical neighbours come from L2).  The backward is a GATHER:
// pixel q collects the contribution of its four neighbouring centres (recomputing their normals) -- no atomics,
// deterministic.  The loss is reduced wave -> block -> a fixed-order final pass in double (deterministic as well).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/radegs.h"

namespace rgn {

struct Maps {
  const float* m1; const float* m2;   // depth: [H,W] each; points: [3,H,W] each
  int W, H;
  float kx0, kx2, ky1, ky2;           // K^-1 rows: x' = kx0*(x+.5) + kx2, y' = ky1*(y+.5) + ky2
};

struct f3 { float x, y, z; };
__device__ __forceinline__ f3 sub(f3 a, f3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ f3 cross(f3 a, f3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
__device__ __forceinline__ float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

template <bool POINTS>
__device__ __forceinline__ f3 point_at(const Maps& m, int k, int y, int x) {
  const float* p = k ? m.m2 : m.m1;
  const size_t HW = (size_t)m.H * m.W, i = (size_t)y * m.W + x;
  if constexpr (POINTS) {
    return {p[i], p[HW + i], p[2 * HW + i]};
  } else {
    const float d = p[i];
    return {d * (m.kx0 * ((float)x + 0.5f) + m.kx2), d * (m.ky1 * ((float)y + 0.5f) + m.ky2), d};
  }
}

constexpr float kEps = 1e-12f;  // torch.nn.functional.normalize

// normal of map k at an INTERIOR centre (y,x); also returns the pieces the backward needs
template <bool POINTS>
__device__ __forceinline__ f3 normal_at(const Maps& m, int k, int y, int x, f3& a, f3& b, float& len) {
  a = sub(point_at<POINTS>(m, k, y + 1, x), point_at<POINTS>(m, k, y - 1, x));
  b = sub(point_at<POINTS>(m, k, y, x + 1), point_at<POINTS>(m, k, y, x - 1));
  const f3 v = cross(a, b);
  len = sqrtf(dot(v, v));
  const float inv = 1.0f / fmaxf(len, kEps);
  return {v.x * inv, v.y * inv, v.z * inv};
}

__device__ __forceinline__ bool interior(const Maps& m, int y, int x) { return y >= 1 && y < m.H - 1 && x >= 1 && x < m.W - 1; }

// d<g, N>/da and /db at one centre
template <bool POINTS>
__device__ __forceinline__ void centre_grads(const Maps& m, int k, int y, int x, f3 g, f3& ga, f3& gb) {
  f3 a, b;
  float len;
  const f3 N = normal_at<POINTS>(m, k, y, x, a, b, len);
  const float inv = 1.0f / fmaxf(len, kEps);
  f3 gv;
  if (len > kEps) {
    const float ng = dot(N, g);
    gv = {(g.x - N.x * ng) * inv, (g.y - N.y * ng) * inv, (g.z - N.z * ng) * inv};
  } else {
    gv = {g.x * inv, g.y * inv, g.z * inv};
  }
  ga = cross(b, gv);
  gb = cross(gv, a);
}

// ------------------------------------------------------------------ normal maps, forward ----
template <bool POINTS>
__global__ void __launch_bounds__(256) normals_fwd_kernel(const Maps m, float* __restrict__ out /* [2,3,H,W] */) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= m.W || y >= m.H) return;
  const size_t HW = (size_t)m.H * m.W, i = (size_t)y * m.W + x;
  const bool in = interior(m, y, x);
#pragma unroll
  for (int k = 0; k < 2; k++) {
    f3 N = {0.f, 0.f, 0.f}, a, b;
    float len;
    if (in) N = normal_at<POINTS>(m, k, y, x, a, b, len);
    out[(3 * k + 0) * HW + i] = N.x; out[(3 * k + 1) * HW + i] = N.y; out[(3 * k + 2) * HW + i] = N.z;
  }
}

// Cotangent source: either a [2,3,H,W] tensor (generic) or -w_k * rendered_normal (fused loss)
struct Cot {
  const float* cot;       // generic
  const float* rn;        // loss: rendered normal [3,H,W]
  const float* upstream;  // loss: d(objective)/d(loss), device scalar (may be null = 1)
  float w0, w1;           // loss: (1-r)/(HW), r/(HW)
};

template <bool LOSS>
__device__ __forceinline__ f3 cot_at(const Cot& c, float up, int k, size_t HW, size_t i) {
  if constexpr (LOSS) {
    const float w = -(k ? c.w1 : c.w0) * up;
    return {w * c.rn[i], w * c.rn[HW + i], w * c.rn[2 * HW + i]};
  } else {
    const float* p = c.cot + (size_t)3 * k * HW;
    return {p[i], p[HW + i], p[2 * HW + i]};
  }
}

// ------------------------------------------------------- backward (generic and fused loss) ----
template <bool POINTS, bool LOSS>
__global__ void __launch_bounds__(256) normals_bwd_kernel(const Maps m, const Cot c, float* __restrict__ g1, float* __restrict__ g2,
                                                         float* __restrict__ g_rn /* LOSS only: [3,H,W] */) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  if (x >= m.W || y >= m.H) return;
  const size_t HW = (size_t)m.H * m.W, i = (size_t)y * m.W + x;
  const float up = (LOSS && c.upstream) ? c.upstream[0] : 1.0f;
  f3 grn = {0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < 2; k++) {
    f3 gp = {0.f, 0.f, 0.f}, ga, gb;
    // q = (y,x) is the +y neighbour of centre (y-1,x), the -y neighbour of (y+1,x), +x of (y,x-1), -x of (y,x+1)
    if (interior(m, y - 1, x)) { centre_grads<POINTS>(m, k, y - 1, x, cot_at<LOSS>(c, up, k, HW, i - m.W), ga, gb); gp = {gp.x + ga.x, gp.y + ga.y, gp.z + ga.z}; }
    if (interior(m, y + 1, x)) { centre_grads<POINTS>(m, k, y + 1, x, cot_at<LOSS>(c, up, k, HW, i + m.W), ga, gb); gp = {gp.x - ga.x, gp.y - ga.y, gp.z - ga.z}; }
    if (interior(m, y, x - 1)) { centre_grads<POINTS>(m, k, y, x - 1, cot_at<LOSS>(c, up, k, HW, i - 1), ga, gb); gp = {gp.x + gb.x, gp.y + gb.y, gp.z + gb.z}; }
    if (interior(m, y, x + 1)) { centre_grads<POINTS>(m, k, y, x + 1, cot_at<LOSS>(c, up, k, HW, i + 1), ga, gb); gp = {gp.x - gb.x, gp.y - gb.y, gp.z - gb.z}; }
    float* g = k ? g2 : g1;
    if constexpr (POINTS) {
      g[i] = gp.x; g[HW + i] = gp.y; g[2 * HW + i] = gp.z;
    } else {
      const f3 r = {m.kx0 * ((float)x + 0.5f) + m.kx2, m.ky1 * ((float)y + 0.5f) + m.ky2, 1.0f};
      g[i] = dot(gp, r);
    }
    if constexpr (LOSS) {
      if (interior(m, y, x)) {
        f3 a, b;
        float len;
        const f3 N = normal_at<POINTS>(m, k, y, x, a, b, len);
        const float w = -(k ? c.w1 : c.w0) * up;
        grn = {grn.x + w * N.x, grn.y + w * N.y, grn.z + w * N.z};
      }
    }
  }
  if constexpr (LOSS) { g_rn[i] = grn.x; g_rn[HW + i] = grn.y; g_rn[2 * HW + i] = grn.z; }
}

// ----------------------------------------------------------------------- fused loss, forward ----
template <bool POINTS>
__global__ void __launch_bounds__(256) normal_loss_fwd_kernel(const Maps m, const float* __restrict__ rn, double* __restrict__ partial) {
  const int x = blockIdx.x * 64 + threadIdx.x, y = blockIdx.y * 4 + threadIdx.y;
  float e0 = 0.f, e1 = 0.f;
  if (x < m.W && y < m.H) {
    const size_t HW = (size_t)m.H * m.W, i = (size_t)y * m.W + x;
    e0 = 1.0f; e1 = 1.0f;  // border: N = 0 -> error 1
    if (interior(m, y, x)) {
      const f3 n = {rn[i], rn[HW + i], rn[2 * HW + i]};
      f3 a, b;
      float len;
      e0 = 1.0f - dot(n, normal_at<POINTS>(m, 0, y, x, a, b, len));
      e1 = 1.0f - dot(n, normal_at<POINTS>(m, 1, y, x, a, b, len));
    }
  }
  // wave (64 lanes = one row segment) -> block (4 waves)
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) { e0 += __shfl_xor(e0, o); e1 += __shfl_xor(e1, o); }
  __shared__ float s[4][2];
  if (threadIdx.x == 0) { s[threadIdx.y][0] = e0; s[threadIdx.y][1] = e1; }
  __syncthreads();
  if (threadIdx.x == 0 && threadIdx.y == 0) {
    const size_t b = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    partial[2 * b] = (double)s[0][0] + (double)s[1][0] + (double)s[2][0] + (double)s[3][0];
    partial[2 * b + 1] = (double)s[0][1] + (double)s[1][1] + (double)s[2][1] + (double)s[3][1];
  }
}

__global__ void __launch_bounds__(256) normal_loss_final_kernel(const double* __restrict__ partial, int nblocks, double inv_hw, float r,
                                                               float* __restrict__ loss /* [3]: loss, mean err0, mean err1 */) {
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
    const float m0 = (float)(s0[0] * inv_hw), m1 = (float)(s1[0] * inv_hw);
    loss[0] = (1.0f - r) * m0 + r * m1;
    loss[1] = m0; loss[2] = m1;
  }
}

static thread_local char g_err[256] = "";
static int fail(int code, const char* what, hipError_t e = hipSuccess) {
  if (e != hipSuccess) snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
  else snprintf(g_err, sizeof(g_err), "%s", what);
  return code;
}

static Maps make_maps(const RadegsNormalArgs* A) {
  Maps m;
  m.m1 = A->map1; m.m2 = A->map2; m.W = A->width; m.H = A->height;
  // graphics_utils.py:99-105 builds K^-1 in double and rounds the entries to float32
  const double fx = A->width / (2.0 * tan(A->fovx / 2.0)), fy = A->height / (2.0 * tan(A->fovy / 2.0));
  m.kx0 = (float)(1.0 / fx); m.kx2 = (float)(-A->width / (2.0 * fx));
  m.ky1 = (float)(1.0 / fy); m.ky2 = (float)(-A->height / (2.0 * fy));
  return m;
}

static int check(const RadegsNormalArgs* A) {
  if (!A) return fail(RADEGS_ERR_INVALID_ARG, "null argument");
  if (A->width <= 0 || A->height <= 0) return fail(RADEGS_ERR_INVALID_ARG, "bad image size");
  if (!A->map1 || !A->map2) return fail(RADEGS_ERR_INVALID_ARG, "input maps missing");
  return 0;
}

}  // namespace rgn

using namespace rgn;

extern "C" {

const char* radegs_normals_last_error(void) { return g_err; }

size_t radegs_normal_loss_scratch_bytes(int width, int height) {
  const size_t nb = (size_t)((width + 63) / 64) * ((height + 3) / 4);
  return nb * 2 * sizeof(double);
}

int radegs_normals_forward(const RadegsNormalArgs* A, float* out_normals, void* stream_v) {
  if (int rc = check(A)) return rc;
  if (!out_normals) return fail(RADEGS_ERR_INVALID_ARG, "output missing");
  hipStream_t s = static_cast<hipStream_t>(stream_v);
  const Maps m = make_maps(A);
  const dim3 grid((A->width + 63) / 64, (A->height + 3) / 4), block(64, 4);
  if (A->points) hipLaunchKernelGGL(normals_fwd_kernel<true>, grid, block, 0, s, m, out_normals);
  else hipLaunchKernelGGL(normals_fwd_kernel<false>, grid, block, 0, s, m, out_normals);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(RADEGS_ERR_HIP, "normals_fwd_kernel", e);
}

int radegs_normals_backward(const RadegsNormalArgs* A, const float* grad_normals, float* grad_map1, float* grad_map2, void* stream_v) {
  if (int rc = check(A)) return rc;
  if (!grad_normals || !grad_map1 || !grad_map2) return fail(RADEGS_ERR_INVALID_ARG, "gradient tensors missing");
  hipStream_t s = static_cast<hipStream_t>(stream_v);
  const Maps m = make_maps(A);
  Cot c{};
  c.cot = grad_normals;
  const dim3 grid((A->width + 63) / 64, (A->height + 3) / 4), block(64, 4);
  if (A->points) hipLaunchKernelGGL((normals_bwd_kernel<true, false>), grid, block, 0, s, m, c, grad_map1, grad_map2, nullptr);
  else hipLaunchKernelGGL((normals_bwd_kernel<false, false>), grid, block, 0, s, m, c, grad_map1, grad_map2, nullptr);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(RADEGS_ERR_HIP, "normals_bwd_kernel", e);
}

int radegs_normal_loss_forward(const RadegsNormalArgs* A, const float* rendered_normal, float depth_ratio, void* scratch,
                               float* out_loss3, void* stream_v) {
  if (int rc = check(A)) return rc;
  if (!rendered_normal || !scratch || !out_loss3) return fail(RADEGS_ERR_INVALID_ARG, "tensor missing");
  hipStream_t s = static_cast<hipStream_t>(stream_v);
  const Maps m = make_maps(A);
  const dim3 grid((A->width + 63) / 64, (A->height + 3) / 4), block(64, 4);
  double* partial = static_cast<double*>(scratch);
  if (A->points) hipLaunchKernelGGL(normal_loss_fwd_kernel<true>, grid, block, 0, s, m, rendered_normal, partial);
  else hipLaunchKernelGGL(normal_loss_fwd_kernel<false>, grid, block, 0, s, m, rendered_normal, partial);
  hipLaunchKernelGGL(normal_loss_final_kernel, dim3(1), dim3(256), 0, s, partial, (int)(grid.x * grid.y),
                     1.0 / ((double)A->width * A->height), depth_ratio, out_loss3);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(RADEGS_ERR_HIP, "normal_loss_fwd_kernel", e);
}

int radegs_normal_loss_backward(const RadegsNormalArgs* A, const float* rendered_normal, float depth_ratio, const float* upstream,
                                float* grad_map1, float* grad_map2, float* grad_rendered_normal, void* stream_v) {
  if (int rc = check(A)) return rc;
  if (!rendered_normal || !grad_map1 || !grad_map2 || !grad_rendered_normal) return fail(RADEGS_ERR_INVALID_ARG, "tensor missing");
  hipStream_t s = static_cast<hipStream_t>(stream_v);
  const Maps m = make_maps(A);
  Cot c{};
  c.rn = rendered_normal; c.upstream = upstream;
  const double hw = (double)A->width * A->height;
  c.w0 = (float)((1.0 - (double)depth_ratio) / hw); c.w1 = (float)((double)depth_ratio / hw);
  const dim3 grid((A->width + 63) / 64, (A->height + 3) / 4), block(64, 4);
  if (A->points) hipLaunchKernelGGL((normals_bwd_kernel<true, true>), grid, block, 0, s, m, c, grad_map1, grad_map2, grad_rendered_normal);
  else hipLaunchKernelGGL((normals_bwd_kernel<false, true>), grid, block, 0, s, m, c, grad_map1, grad_map2, grad_rendered_normal);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : fail(RADEGS_ERR_HIP, "normals_bwd_kernel", e);
}

}  // extern "C"
