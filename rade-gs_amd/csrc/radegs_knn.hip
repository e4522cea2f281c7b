// radegs_knn.hip -- simple_knn._C.distCUDA2: for every point, the mean of the squared distances to its 3 nearest neighbours
// (scene/gaussian_model.py:315 initialises the Gaussian scales from it; one call per training run).  The reference imports
// it from the `simple-knn` submodule (gitlab.inria.fr/bkerbl/simple-knn), whose source is NOT vendored under /root/reference
// (empty directory, .gitmodules:1-3), so this restates the published algorithm of that library:
//   Morton-order the points (10 bits per axis over the bounding box), cut the order into boxes of 1024 points with their
//   AABBs, seed each point's 3-best list from its +-3 neighbours in Morton order, then visit every box whose AABB is closer
//   than the current 3rd-best distance and scan it.  Exact 3-NN (the seed only provides the rejection radius).
// gfx950 notes: the sort is this library's own LSD radix sort (radegs_sort.hip, 30-bit keys); box AABBs are walked from
// LDS by a whole wave at a time (wave-uniform loop over boxes, per-lane rejection), points are gathered in Morton order so
// the lanes of a wave sit close in space and take the same boxes.
#include <hip/hip_runtime.h>
#include <float.h>
#include <stdint.h>

#include "../../include/radegs.h"
#include "rg_prims.h"

namespace rgk {

constexpr int kBox = 1024;

__device__ __forceinline__ uint32_t spread10(uint32_t x) {  // 10 bits -> every third bit
  x = (x | (x << 16)) & 0x030000FF;
  x = (x | (x << 8)) & 0x0300F00F;
  x = (x | (x << 4)) & 0x030C30C3;
  x = (x | (x << 2)) & 0x09249249;
  return x;
}

// order-preserving float <-> uint for atomic min/max
__device__ __forceinline__ uint32_t f2ord(float f) { const uint32_t u = __float_as_uint(f); return (u & 0x80000000u) ? ~u : (u | 0x80000000u); }
__device__ __forceinline__ float ord2f(uint32_t u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7FFFFFFFu) : ~u); }

__global__ void __launch_bounds__(256) bounds_kernel(int P, const float* __restrict__ pts, uint32_t* __restrict__ mm /* [6]: min xyz, max xyz */) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  if (i < P) {
#pragma unroll
    for (int c = 0; c < 3; c++) lo[c] = hi[c] = pts[3 * (size_t)i + c];
  }
#pragma unroll
  for (int c = 0; c < 3; c++) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo[c] = fminf(lo[c], __shfl_xor(lo[c], o)); hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o)); }
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int c = 0; c < 3; c++) { atomicMin(&mm[c], f2ord(lo[c])); atomicMax(&mm[3 + c], f2ord(hi[c])); }
  }
}

__global__ void __launch_bounds__(256) morton_kernel(int P, const float* __restrict__ pts, const uint32_t* __restrict__ mm, uint32_t* __restrict__ codes) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P) return;
  uint32_t code = 0;
#pragma unroll
  for (int c = 0; c < 3; c++) {
    const float lo = ord2f(mm[c]), hi = ord2f(mm[3 + c]);
    const float ext = hi - lo;
    const float u = ext > 0.f ? (pts[3 * (size_t)i + c] - lo) / ext : 0.f;
    const uint32_t q = (uint32_t)fminf(fmaxf(u * 1023.0f, 0.f), 1023.f);
    code |= spread10(q) << (2 - c);
  }
  codes[i] = code;
}

// sorted copy of the points + AABB of every box of kBox consecutive sorted points
__global__ void __launch_bounds__(256) gather_boxes_kernel(int P, const float* __restrict__ pts, const uint32_t* __restrict__ order,
                                                          float* __restrict__ sorted /* [P][3] */, float* __restrict__ boxes /* [nb][6] */) {
  __shared__ float red[4][6];
  const int b = blockIdx.x;
  float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (int k = threadIdx.x; k < kBox; k += 256) {
    const int i = b * kBox + k;
    if (i < P) {
      const size_t src = order[i];
#pragma unroll
      for (int c = 0; c < 3; c++) {
        const float v = pts[3 * src + c];
        sorted[3 * (size_t)i + c] = v;
        lo[c] = fminf(lo[c], v); hi[c] = fmaxf(hi[c], v);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 3; c++) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { lo[c] = fminf(lo[c], __shfl_xor(lo[c], o)); hi[c] = fmaxf(hi[c], __shfl_xor(hi[c], o)); }
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int c = 0; c < 3; c++) { red[wave][c] = lo[c]; red[wave][3 + c] = hi[c]; }
  }
  __syncthreads();
  if (threadIdx.x < 6) {
    const int c = threadIdx.x;
    float v = red[0][c];
    for (int w = 1; w < 4; w++) v = c < 3 ? fminf(v, red[w][c]) : fmaxf(v, red[w][c]);
    boxes[6 * (size_t)b + c] = v;
  }
}

__device__ __forceinline__ void update3(float d, float best[3]) {
  if (d < best[2]) {
    if (d < best[1]) {
      best[2] = best[1];
      if (d < best[0]) { best[1] = best[0]; best[0] = d; } else best[1] = d;
    } else best[2] = d;
  }
}
__device__ __forceinline__ float dist2(float ax, float ay, float az, float bx, float by, float bz) {
  const float dx = ax - bx, dy = ay - by, dz = az - bz;
  return dx * dx + dy * dy + dz * dz;
}

__global__ void __launch_bounds__(64) knn3_kernel(int P, int nb, const float* __restrict__ sorted, const float* __restrict__ boxes,
                                                 const uint32_t* __restrict__ order, float* __restrict__ out) {
  const int i = blockIdx.x * 64 + threadIdx.x;
  const bool live = i < P;
  const int ic = live ? i : P - 1;
  const float px = sorted[3 * (size_t)ic], py = sorted[3 * (size_t)ic + 1], pz = sorted[3 * (size_t)ic + 2];
  float best[3] = {FLT_MAX, FLT_MAX, FLT_MAX};
  // seed: +-3 neighbours in Morton order -> an upper bound of the 3rd-NN distance
  for (int k = max(0, ic - 3); k <= min(P - 1, ic + 3); k++)
    if (k != ic) update3(dist2(px, py, pz, sorted[3 * (size_t)k], sorted[3 * (size_t)k + 1], sorted[3 * (size_t)k + 2]), best);
  const float reject = best[2];
  best[0] = best[1] = best[2] = FLT_MAX;
  for (int b = 0; b < nb; b++) {
    const float* bx = boxes + 6 * (size_t)b;  // wave-uniform address: scalar loads
    const float dx = fmaxf(fmaxf(bx[0] - px, px - bx[3]), 0.f), dy = fmaxf(fmaxf(bx[1] - py, py - bx[4]), 0.f),
                dz = fmaxf(fmaxf(bx[2] - pz, pz - bx[5]), 0.f);
    const float d = dx * dx + dy * dy + dz * dz;
    const bool want = live && !(d > reject || d > best[2]);
    if (!__any(want)) continue;
    const int k0 = b * kBox, k1 = min(P, k0 + kBox);
    for (int k = k0; k < k1; k++) {  // wave-uniform walk; lanes that rejected the box just do not update
      const float q = dist2(px, py, pz, sorted[3 * (size_t)k], sorted[3 * (size_t)k + 1], sorted[3 * (size_t)k + 2]);
      if (want && k != ic) update3(q, best);
    }
  }
  if (live) out[order[i]] = (best[0] + best[1] + best[2]) / 3.0f;
}

}  // namespace rgk

extern "C" {

size_t radegs_knn_scratch_bytes(int P) {
  const size_t n = (size_t)(P > 0 ? P : 1), nb = (n + rgk::kBox - 1) / rgk::kBox;
  return 256 /* bounds */ + 4 * (n * 4 + 256) /* codes, sorted codes, order */ + n * 12 + 256 + nb * 24 + 256 + rg::sort_temp_bytes(n) + 1024;
}

int radegs_knn_mean_dist2(int P, const float* points, void* scratch, float* out, void* stream_v) {
  if (P < 0) return RADEGS_ERR_INVALID_ARG;
  if (P == 0) return 0;
  if (!points || !scratch || !out) return RADEGS_ERR_INVALID_ARG;
  hipStream_t s = static_cast<hipStream_t>(stream_v);
  char* p = static_cast<char*>(scratch);
  auto take = [&](size_t bytes) { char* r = p; p += (bytes + 255) & ~size_t(255); return r; };
  const size_t n = (size_t)P, nb = (n + rgk::kBox - 1) / rgk::kBox;
  uint32_t* mm = reinterpret_cast<uint32_t*>(take(6 * 4));
  uint32_t* codes = reinterpret_cast<uint32_t*>(take(n * 4));
  uint32_t* codes_sorted = reinterpret_cast<uint32_t*>(take(n * 4));
  uint32_t* order = reinterpret_cast<uint32_t*>(take(n * 4));
  float* sorted = reinterpret_cast<float*>(take(n * 12));
  float* boxes = reinterpret_cast<float*>(take(nb * 24));
  char* stemp = take(rg::sort_temp_bytes(n));
  const uint32_t init[6] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u};
  if (hipMemcpyAsync(mm, init, sizeof(init), hipMemcpyHostToDevice, s) != hipSuccess) return RADEGS_ERR_HIP;
  hipLaunchKernelGGL(rgk::bounds_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, points, mm);
  hipLaunchKernelGGL(rgk::morton_kernel, dim3((P + 255) / 256), dim3(256), 0, s, P, points, mm, codes);
  if (rg::radix_sort_pairs_u32(stemp, rg::sort_temp_bytes(n), codes, codes_sorted, nullptr, order, n, 30, s) != hipSuccess) return RADEGS_ERR_HIP;
  hipLaunchKernelGGL(rgk::gather_boxes_kernel, dim3((unsigned)nb), dim3(256), 0, s, P, points, order, sorted, boxes);
  hipLaunchKernelGGL(rgk::knn3_kernel, dim3((P + 63) / 64), dim3(64), 0, s, P, (int)nb, sorted, boxes, order, out);
  return hipGetLastError() == hipSuccess ? 0 : RADEGS_ERR_HIP;
}

}  // extern "C"
