// radegs_kernels.hip -- gfx950 kernels of the differentiable splat rasterizer.
//
// Stage map (reference kernel -> this file); DGR = submodules/diff-gaussian-rasterization:
//   preprocessCUDA<3,false>   DGR/cuda_rasterizer/forward.cu:307-423        -> preprocess_fwd_kernel
//   duplicateWithKeys         DGR/cuda_rasterizer/rasterizer_impl.cu:70-111 -> emit_instances_kernel
//   identifyTileRanges        rasterizer_impl.cu:151-173                    -> tile_ranges_kernel
//   renderCUDA fwd            forward.cu:428-693                            -> blend_fwd_kernel
//   renderCUDA bwd            DGR/cuda_rasterizer/backward.cu:631-1016      -> blend_bwd_kernel
//   computeCov2DCUDA + preprocessCUDA bwd  backward.cu:145-488,560-628      -> preprocess_bwd_kernel
//   checkFrustum              rasterizer_impl.cu:54-66                      -> mark_visible_kernel
//
// Design for CDNA4 (not a translation of the CUDA block structure):
//   * Blend kernels: ONE wave64 owns a 16 x (4*PPL) pixel strip of a 16x16 tile; each lane keeps
//     PPL pixels (same column, rows 4 apart) in registers.  No cross-wave sharing, no barriers in
//     the hot loop.  Per-Gaussian attributes are wave-uniform: they are staged 64 entries at a
//     time into wave-private LDS (each lane gathers one 64-B record = one cache line) and read
//     back as broadcast ds_read_b128 -- amortised over PPL pixels per lane, which keeps the
//     LDS pipe (one b128 per 4 clk per CU) off the critical path.
//   * The skip test needs no transcendental: a per-Gaussian exponent threshold (computed once in
//     preprocess) rejects alpha < 1/255 pairs from the quadratic form alone; exp is evaluated
//     only for surviving pairs, with the specified exp_spec() so that every thresholded decision
//     matches the CPU oracle bit-for-bit.
//   * Backward: per-lane partial gradients of one Gaussian are reduced with a butterfly that
//     reduces all 16 (or 32) gradient components at once (15 + 2 cross-lane exchanges instead of
//     16 x 6) and leaves component c in lane c, so ONE 16/25-lane global_atomic_add_f32
//     instruction updates one 64-B accumulator line.
//   * blockIdx -> tile mapping gives each XCD a contiguous band of tiles (neighbouring tiles
//     share splat records -> per-XCD L2 reuse).
// No MFMA: there is no dense contraction on this path.
#include <hip/hip_runtime.h>
#include <type_traits>

#include "rg_blend.h"
#include "rg_layout.h"
#include "rg_preprocess.h"
#include "rg_preprocess_bwd.h"

namespace rg {

// ------------------------------------------------------------------ launch arguments ----
struct CamArgs {
  const float* view;    // device [16]
  const float* proj;    // device [16]
  const float* campos;  // device [3]
  float focal_x, focal_y, tan_fovx, tan_fovy, kernel_size, scale_modifier;
  int W, H, gx, gy;
};

__device__ __forceinline__ Camera load_camera(const CamArgs& a) {
  Camera c;
#pragma unroll
  for (int i = 0; i < 16; i++) { c.view[i] = a.view[i]; c.proj[i] = a.proj[i]; }
#pragma unroll
  for (int i = 0; i < 3; i++) c.campos[i] = a.campos[i];
  c.focal_x = a.focal_x; c.focal_y = a.focal_y; c.tan_fovx = a.tan_fovx; c.tan_fovy = a.tan_fovy;
  c.kernel_size = a.kernel_size; c.scale_modifier = a.scale_modifier;
  c.W = a.W; c.H = a.H; c.gx = a.gx; c.gy = a.gy;
  return c;
}

// =========================================================================== preprocess ==
constexpr uint32_t kDepthKeyBase = 0x3E4CCCCDu;   // bits(0.2f): every visible Gaussian lies beyond the near plane (auxiliary.h:166)
struct PreFwdArgs {
  int P, D, M;
  const float* means3D; const float* scales; const float* rotations; const float* cov3D_precomp;
  const float* opacities; const float* shs; const float* colors_precomp;
  CamArgs cam;
  int write_b;
  int* radii; float4* splat_a; float4* splat_b; uint32_t* tiles_touched; uint32_t* depth_key; uint8_t* clamped;
  float4* inte_rec;  // [P][2] {icr0..icr3 | icr4, icr5, well, 0}; INTE kernel only
  uint32_t* rect;    // [P] packed tile rectangle
  uint32_t* key_overflow;   // mapped host word (or null): set when a visible depth key does not fit the 27-bit window of the 3-pass sort
};

template <bool INTE>
__global__ void __launch_bounds__(256) preprocess_fwd_kernel(const PreFwdArgs a) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.P) return;
  const Camera cam = load_camera(a.cam);
  SplatFwd s;
  const float* m = a.means3D + 3 * (size_t)idx;
  const v3 p_orig = mk3(m[0], m[1], m[2]);
  const float* sh = a.shs ? a.shs + (size_t)idx * a.M * 3 : nullptr;
  const float* color_in = a.colors_precomp ? a.colors_precomp + 3 * (size_t)idx : nullptr;
  const float* scale3 = a.scales ? a.scales + 3 * (size_t)idx : nullptr;
  const float* quat4 = a.rotations ? a.rotations + 4 * (size_t)idx : nullptr;
  const float* cov_in = a.cov3D_precomp ? a.cov3D_precomp + 6 * (size_t)idx : nullptr;
  preprocess_fwd<INTE>(p_orig, scale3, quat4, cov_in, a.opacities[idx], a.D, sh, color_in, cam, s);
  a.radii[idx] = s.radius;
  a.tiles_touched[idx] = (uint32_t)s.tiles;
  a.rect[idx] = s.radius > 0 ? s.rect : 0u;
  // positive floats order like unsigned ints; invisible Gaussians sort to the very end
  a.depth_key[idx] = s.radius > 0 ? __float_as_uint(s.depth) : 0xFFFFFFFFu;
  // the depth sort runs three 9-bit passes over (key - bits(0.2f)) when every visible key fits 27 bits, i.e. z < 13 107 (rg_launch.inc);
  // a key outside raises the flag the host looks at before it trusts the order (and redoes the forward with the 4-pass sort)
  if (a.key_overflow && s.radius > 0 && __float_as_uint(s.depth) - kDepthKeyBase >= (1u << 27))
    __hip_atomic_store(a.key_overflow, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (s.radius > 0) {
    float4* ra = a.splat_a + 4 * (size_t)idx;
    ra[0] = make_float4(s.mx, s.my, s.cx, s.cy);
    ra[1] = make_float4(s.cz, s.op, skip_threshold(s.op), s.ts);
    ra[2] = make_float4(s.rgb[0], s.rgb[1], s.rgb[2], s.rp[0]);
    ra[3] = make_float4(s.rp[1], s.nrm[0], s.nrm[1], s.nrm[2]);
    if (a.write_b) {
      float4* rb = a.splat_b + 3 * (size_t)idx;
      rb[0] = make_float4(s.cp[0], s.cp[1], s.cp[2], s.cp[3]);
      rb[1] = make_float4(s.cp[4], s.cp[5], s.vp[0], s.vp[1]);
      rb[2] = make_float4(s.vp[2], 0.f, 0.f, 0.f);
    }
    a.clamped[idx] = (uint8_t)s.clamped;   // bits 0..2: SH clamp flags
    if constexpr (INTE) {
      float4* ri = a.inte_rec + 2 * (size_t)idx;
      ri[0] = make_float4(s.icr[0], s.icr[1], s.icr[2], s.icr[3]);
      ri[1] = make_float4(s.icr[4], s.icr[5], s.well ? 1.0f : 0.0f, 0.f);
    }
  }
}

__global__ void __launch_bounds__(256) mark_visible_kernel(int P, const float* means3D, const float* view, unsigned char* present) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= P) return;
  const float* m = means3D + 3 * (size_t)idx;
  v3 pv = xform43(mk3(m[0], m[1], m[2]), view);
  present[idx] = !(pv.z <= 0.2f);
}

// ============================================================================== binning ==
// One thread per Gaussian IN DEPTH ORDER; writes (tile id, gaussian idx) for every tile of its rect, rows outer /
// columns inner -- the emission order of rasterizer_impl.cu:98-109.  Splats with few tiles are written by their own
// lane; a splat with many tiles (heavy-overdraw scenes: hundreds per splat) is handed to the whole wave, which writes
// its instances 64 at a time to consecutive addresses -- coalesced, and no lane serialises a 300-iteration loop.
constexpr int kEmitCoopThreshold = 16;
// rect != nullptr (tile grid at most 255x255): rect[i] is the packed tile rectangle of the i-th Gaussian IN DEPTH ORDER (the scan's
// gather wrote it): no random access at all here, instead of recomputing the rectangle from three gathers.
// MASKS (sub-tile entry streams, rg_streams.inc): the instance value also carries, in its top byte, which of the tile's eight 8x4
// blocks the splat can reach (ellipse_tile_mask, rg_blend.h).  Here the splat's record is in registers once for all its tiles; after
// the sort the same question costs a dependent gather per list entry.  block_lists_kernel strips the byte again.
template <bool MASKS>
__global__ void __launch_bounds__(256) emit_instances_kernel(int P, const uint32_t* idx_sorted, const uint32_t* offsets,
                                                            const uint32_t* tiles_touched, const float4* splat_a, const int* radii,
                                                            const uint32_t* rect, int gx, int gy, uint32_t* tile_keys, uint32_t* vals,
                                                            uint32_t cap, uint32_t* ranges_to_clear, uint32_t* count_mirror, uint32_t seq, uint32_t* stream_tag, int key16,
                                                            int mask_in_key, const unsigned long long* tile_sq_sum) {
  // key16: tile ids leave as 16-bit keys (grids of at most 65 536 tiles): the tile sort then moves a third less (radegs_sort.hip)
  // mask_in_key (MASKS, scenes of 2^24 Gaussians and more, 32-bit keys): the block mask rides in the top byte of the KEY -- the tile sort
  // only looks at the low tile bits -- and the value is the plain Gaussian index
  uint16_t* const tile_keys16 = reinterpret_cast<uint16_t*>(tile_keys);
  // cap: capacity of tile_keys/vals.  With exact allocation it equals num_rendered; in the speculative path (rg_launch.inc)
  // it is a prediction and instances beyond it are dropped here (the host detects the overflow and redoes the binning).
  const int i = blockIdx.x * 256 + threadIdx.x;
  // Two chores that used to be kernels of their own (a 5 us copy and a 5 us fill per forward): the tile ranges start at (0,0)
  // (rasterizer_impl.cu:383's memset; tile_ranges_kernel runs two sorts later), and num_rendered goes to the host's pinned words
  // without a copy engine command or an event: the count, then this forward's sequence number with release order (the host polls the
  // sequence word once everything else of the forward is queued, rg_launch.inc::binning_finish).
  if (ranges_to_clear) {
    for (int k = i; k < 2 * gx * gy; k += (int)gridDim.x * 256) ranges_to_clear[k] = 0u;
  }
  if (count_mirror && i == 0) {
    __hip_atomic_store(count_mirror, offsets[P - 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const unsigned long long sq = tile_sq_sum ? *tile_sq_sum : 0ull;   // PIN_SQ_LO / PIN_SQ_HI (rg_launch.inc)
    __hip_atomic_store(count_mirror + 6, (uint32_t)sq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    __hip_atomic_store(count_mirror + 7, (uint32_t)(sq >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    // PIN_SEQ; a stream forward publishes it later, together with the chunks its lists took (block_lists_kernel)
    if (!MASKS) __hip_atomic_store(count_mirror + 2, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (i == 0) {   // does this image state hold entry streams?  (ImageState::stream_tag; the chunk allocator and its overflow flag start at 0)
    stream_tag[0] = MASKS ? kStreamTag : 0u; stream_tag[1] = 0u; stream_tag[3] = 0u;
  }
  const int lane = threadIdx.x & 63;
  uint32_t idx = 0, ntiles = 0, off = 0;
  int x0 = 0, y0 = 0, x1 = 0, y1 = 0;
  float4 q0 = make_float4(0.f, 0.f, 0.f, 0.f), q1 = q0;
  if (i < P) {
    idx = idx_sorted[i];
    if (rect) {
      const uint32_t r = rect[i];   // already in depth order (the scan gathered it)
      x0 = (int)(r & 255u); y0 = (int)((r >> 8) & 255u); x1 = x0 + (int)((r >> 16) & 255u); y1 = y0 + (int)(r >> 24);
      ntiles = (uint32_t)((x1 - x0) * (y1 - y0));
      if (ntiles) off = (i == 0) ? 0u : offsets[i - 1];
      if (MASKS && ntiles) { q0 = splat_a[4 * (size_t)idx]; q1 = splat_a[4 * (size_t)idx + 1]; }
    } else {
      ntiles = tiles_touched[idx];
      if (ntiles) {
        off = (i == 0) ? 0u : offsets[i - 1];
        q0 = splat_a[4 * (size_t)idx];
        if (MASKS) q1 = splat_a[4 * (size_t)idx + 1];
        tile_rect(q0.x, q0.y, radii[idx], gx, gy, x0, y0, x1, y1);
      }
    }
  }
  if constexpr (MASKS) {
    // Entry streams: every INSTANCE gets a block mask (~150 instructions), and a lane that walks its own splat's tiles leaves most of
    // the wave idle (a wave costs its LARGEST splat: 2x2 tiles next to 4x4).  The wave therefore expands its 64 splats into their
    // instances and deals those to the lanes 64 at a time: exclusive prefix of the tile counts, lane t of a chunk finds its splat by a
    // binary search over the prefixes (6 ds_bpermute steps), fetches that splat's rectangle and its ellipse set-up (computed once, by
    // the owning lane, for the whole rectangle) and evaluates ONE tile: 4 slabs, 2 columns each.  Same instances at the same positions
    // (offset of the splat + row-major position in its rectangle).
    uint32_t incl = ntiles;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t y = __shfl_up(incl, d);
      if (lane >= d) incl += y;
    }
    const uint32_t excl = incl - ntiles;
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    const int w_own = x1 - x0;
    EllipseSetup e_own;
    e_own.kind = 0; e_own.cy = 0.f; e_own.det = 0.f; e_own.cxM = 0.f; e_own.icx = 0.f; e_own.hx = 0.f; e_own.hy = 0.f; e_own.tstar = 0.f; e_own.eps = 0.f;
    if (ntiles) {
      const float U = fmaxf(fabsf((float)(x0 * 16) - q0.x), fabsf((float)(x1 * 16 - 1) - q0.x));
      const float V = fmaxf(fabsf((float)(y0 * 16) - q0.y), fabsf((float)(y1 * 16 - 1) - q0.y));
      e_own = ellipse_setup(q0.x, q0.y, q0.z, q0.w, q1.x, q1.z, U, V);
    }
    for (uint32_t base = 0; base < total; base += 64) {
      const uint32_t t = base + (uint32_t)lane;
      int g = 0;                                   // largest lane k with excl_k <= t (excl is non-decreasing, excl_0 = 0)
#pragma unroll
      for (int step = 32; step > 0; step >>= 1) {
        const uint32_t pm = __shfl(excl, g + step);
        if (pm <= t) g += step;
      }
      const uint32_t local = t - __shfl(excl, g);
      const uint32_t g_idx = __shfl(idx, g), g_off = __shfl(off, g);
      const int g_x0 = __shfl(x0, g), g_y0 = __shfl(y0, g), g_w = __shfl(w_own, g);
      const float g_mx = __shfl(q0.x, g), g_my = __shfl(q0.y, g);
      EllipseSetup e;
      e.kind = __shfl(e_own.kind, g); e.cy = __shfl(e_own.cy, g); e.det = __shfl(e_own.det, g); e.cxM = __shfl(e_own.cxM, g);
      e.icx = __shfl(e_own.icx, g); e.hx = __shfl(e_own.hx, g); e.hy = __shfl(e_own.hy, g); e.tstar = __shfl(e_own.tstar, g);
      e.eps = __shfl(e_own.eps, g);
      if (t < total) {
        // row-major position in the splat's rectangle: local = ty * w + tx (local < 2^24: exact in float; the quotient is corrected by one)
        uint32_t ty = (uint32_t)((float)local * __builtin_amdgcn_rcpf((float)g_w));
        int tx = (int)local - (int)(ty * (uint32_t)g_w);
        if (tx < 0) { ty--; tx += g_w; } else if (tx >= g_w) { ty++; tx -= g_w; }
        const uint32_t pos = g_off + local;
        if (pos < cap) {
          const uint32_t mask = ellipse_tile_mask(e, (float)((g_x0 + tx) * 16) - g_mx, (float)((g_y0 + (int)ty) * 16) - g_my) << kMaskShift;
          const uint32_t tile = (uint32_t)((g_y0 + (int)ty) * gx + (g_x0 + tx));
          if (key16) tile_keys16[pos] = (uint16_t)tile; else tile_keys[pos] = tile | (mask_in_key ? mask : 0u);
          vals[pos] = mask_in_key ? g_idx : (g_idx | mask);
        }
      }
    }
    return;
  }
  // tile-wide kernels: no masks.  Splats with few tiles are written by their own lane; a splat with many tiles (heavy-overdraw scenes:
  // hundreds per splat) is handed to the whole wave, which writes its instances 64 at a time to consecutive addresses.
  const bool big = ntiles > (uint32_t)kEmitCoopThreshold;
  if (ntiles && !big) {
    for (int y = y0; y < y1; y++) {
      for (int x = x0; x < x1; x++) {
        if (off < cap) {
          if (key16) tile_keys16[off] = (uint16_t)(y * gx + x); else tile_keys[off] = (uint32_t)(y * gx + x);
          vals[off] = idx;
        }
        off++;
      }
    }
  }
  uint64_t todo = __ballot(big);
  while (todo) {
    const int src = __builtin_ctzll(todo);
    todo &= todo - 1;
    const uint32_t g_idx = __shfl(idx, src), g_n = __shfl(ntiles, src), g_off = __shfl(off, src);
    const int g_x0 = __shfl(x0, src), g_y0 = __shfl(y0, src), g_w = __shfl(x1, src) - g_x0;
    for (uint32_t t = lane; t < g_n; t += 64) {
      const int ty = (int)(t / (uint32_t)g_w), tx = (int)(t - (uint32_t)ty * (uint32_t)g_w);
      if (g_off + t < cap) {
        if (key16) tile_keys16[g_off + t] = (uint16_t)((g_y0 + ty) * gx + (g_x0 + tx)); else tile_keys[g_off + t] = (uint32_t)((g_y0 + ty) * gx + (g_x0 + tx));
        vals[g_off + t] = g_idx;
      }
    }
  }
}

// key_mask: the bits of a key that are the tile id (32-bit keys of a scene of 2^24 Gaussians and more carry the block mask above them)
// One 16-byte load of consecutive keys per thread (8 of the 16-bit ones) and the key before them (rounds 1-5: one key and its predecessor
// per thread, two 2-byte loads: 107 us for C5's 50 M keys); `cap` = items the buffer holds (a vector that would reach past it goes key by key).
constexpr int kRangeThreads = 256;
template <class K>
__global__ void __launch_bounds__(kRangeThreads) tile_ranges_kernel(int L, const K* keys, uint2* ranges, const uint32_t* L_dev, uint32_t key_mask, int cap) {
  constexpr int V = 16 / (int)sizeof(K);
  if (L_dev) L = (int)min((uint32_t)L, *L_dev);  // capacity launch, see emit_instances_kernel
  const long long base = ((long long)blockIdx.x * kRangeThreads + threadIdx.x) * V;
  if (base >= L) return;
  K v[V];
  if (base + V <= cap) {
    const uint4 w = *reinterpret_cast<const uint4*>(keys + base);
    __builtin_memcpy(v, &w, 16);
  } else {
#pragma unroll
    for (int k = 0; k < V; k++) v[k] = base + k < cap ? keys[base + k] : (K)0;
  }
  uint32_t prev = base > 0 ? ((uint32_t)keys[base - 1] & key_mask) : 0u;
#pragma unroll
  for (int k = 0; k < V; k++) {
    const long long i = base + k;
    if (i >= L) break;
    const uint32_t cur = (uint32_t)v[k] & key_mask;
    if (i == 0) ranges[cur].x = 0;
    else if (cur != prev) { ranges[prev].y = (uint32_t)i; ranges[cur].x = (uint32_t)i; }
    if (i == L - 1) ranges[cur].y = (uint32_t)L;
    prev = cur;
  }
}

// Batch-level culling.  While a batch of 64 list entries is staged, lane k still has entry k's
// record in registers; it tests the axis-aligned bounding box of the entry's {power >= thr} ellipse
// (the only region where alpha can reach 1/255) against the pixel rectangle this wave owns.  One
// ballot then gives the 64-bit set of entries worth visiting; the serial walk only touches those.
// The ellipse  1/2 (cx dx^2 + 2 cy dx dy + cz dy^2) <= -thr  has half extents
// hx = sqrt(m cz / det), hy = sqrt(m cx / det), m = -2 thr, det = cx cz - cy^2.  They are inflated by
// 0.1 % + 0.01 px so fp32 rounding of `power` at a pixel can never disagree; any NaN makes the
// comparisons false, i.e. keeps the entry (conservative).  thr > 0 (opacity so low that even the
// centre fails) means the entry can never blend.
__device__ __forceinline__ bool entry_may_touch(const float4 q0, const float4 q1, float x_lo, float x_hi, float y_lo, float y_hi) {
  const float mx = q0.x, my = q0.y, cx = q0.z, cy = q0.w, cz = q1.x, thr = q1.z;
  const float det = cx * cz - cy * cy;
  const float m = -2.0f * thr;
  const float r = m / det;
  const float hx = sqrtf(r * cz) * 1.001f + 0.01f;
  const float hy = sqrtf(r * cx) * 1.001f + 0.01f;
  const bool off = (mx + hx < x_lo) || (mx - hx > x_hi) || (my + hy < y_lo) || (my - hy > y_hi);
  // The extents only mean something for a positive-definite conic.  A needle-thin splat's determinant can round to <= 0 (the
  // preprocess only rejects det == 0, forward.cu:127): such an entry is kept and the exact per-pixel rule decides, as for NaN.
  const bool definite = (det > 0.0f) && (cx > 0.0f) && (cz > 0.0f);
  return !(thr > 0.0f) && !(off && definite);
}

__device__ __forceinline__ int xcd_band_remap(int b, int n);

// ============================================================================ integrate ==
// GaussianRasterizer.integrate (GOF-style point integration used by mesh extraction): for every query point that
// projects into the image, the opacity accumulated along its pixel's ray up to the point.
//   preprocessPointsCUDA  DGR/cuda_rasterizer/forward.cu:855-900   -> points_preprocess_kernel
//   createWithKeys + SortPairs + identifyTileRanges (rasterizer_impl.cu:114-145,784-806)
//                                                                   -> per-PIXEL counting sort (count / scan / scatter)
//   integrateCUDA         forward.cu:938-1372                       -> integrate_kernel
// Re-design: the reference bins points per 16x16 tile and lets every pixel thread scan its tile's whole point list to
// find its own points (two per-thread local arrays of 2048 + 5x256 entries).  A point belongs to exactly one pixel
// (floor of its projection) and points do not interact, so they are binned per pixel here: each lane gets the [start,end)
// range of its own points, the 2048-entry "contributed" list is replaced by replaying the 5-sample transmittance test
// (identical arithmetic => identical decisions), and no per-thread scratch arrays exist at all.
struct PointsPreArgs {
  int PN; const float* points3D; const float* view; float focal_x, focal_y; int W, H;
  float2* p2d; float* pdepth; uint32_t* ppix; uint32_t* pix_count;
  float* out_alpha_integrated; float* out_color_integrated; float* out_coordinate2d; float* out_sdf;
};

// initial values of rasterize_points.cu:312-320
__device__ __forceinline__ void point_outputs_init(const PointsPreArgs& a, int i) {
  a.out_alpha_integrated[i] = 1.0f;
  a.out_color_integrated[3 * (size_t)i] = 0.f; a.out_color_integrated[3 * (size_t)i + 1] = 0.f; a.out_color_integrated[3 * (size_t)i + 2] = 0.f;
  a.out_coordinate2d[2 * (size_t)i] = 0.f; a.out_coordinate2d[2 * (size_t)i + 1] = 0.f;
  a.out_sdf[i] = -1000.0f;
}
__global__ void __launch_bounds__(256) points_init_kernel(const PointsPreArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < a.PN) point_outputs_init(a, i);
}

__global__ void __launch_bounds__(256) points_preprocess_kernel(const PointsPreArgs a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= a.PN) return;
  point_outputs_init(a, i);
  a.ppix[i] = 0xFFFFFFFFu;
  const float* p = a.points3D + 3 * (size_t)i;
  const v3 pv = xform43(mk3(p[0], p[1], p[2]), a.view);
  if (pv.z <= 0.2f) return;
  const float ix = (float)((double)(a.focal_x * pv.x / (pv.z + 0.0000001f)) + a.W / 2.);
  const float iy = (float)((double)(a.focal_y * pv.y / (pv.z + 0.0000001f)) + a.H / 2.);
  if (ix < 0 || ix >= a.W || iy < 0 || iy >= a.H) return;
  a.pdepth[i] = sqrtf(pv.x * pv.x + pv.y * pv.y + pv.z * pv.z);
  a.p2d[i] = make_float2(ix, iy);
  const uint32_t pix = (uint32_t)f2i_sat(floorf(iy)) * (uint32_t)a.W + (uint32_t)f2i_sat(floorf(ix));
  a.ppix[i] = pix;
  atomicAdd(&a.pix_count[pix], 1u);
}

// slot = incl[pix] - (old remaining count): distinct slots inside the pixel's range, no second cursor array
__global__ void __launch_bounds__(256) points_scatter_kernel(int PN, const uint32_t* ppix, uint32_t* pix_count, const uint32_t* pix_incl,
                                                            uint32_t* pt_sorted) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= PN) return;
  const uint32_t pix = ppix[i];
  if (pix == 0xFFFFFFFFu) return;
  const uint32_t rem = atomicSub(&pix_count[pix], 1u);
  pt_sorted[pix_incl[pix] - rem] = (uint32_t)i;
}

struct IntegrateArgs {
  const uint2* ranges; const uint32_t* point_list; const float4* splat_a; const float4* inte_rec;
  int W, H, gx; const float* bg;
  const uint32_t* pix_incl; const uint32_t* pt_sorted; const float2* p2d; const float* pdepth;
  float* out9; float* final_T; uint32_t* n_contrib;
  float* out_alpha_integrated; float* out_color_integrated; float* out_coordinate2d; float* out_sdf;
};

constexpr int kMaxContributors = 512 * 4;  // MAX_NUM_CONTRIBUTORS * 4, auxiliary.h:27 / forward.cu:1003
constexpr int kPointsPerPass = 4;

// The 5-sample (centre + 4 corners) transmittance test of forward.cu:1043-1110 for one list entry; updates cT and
// reports which samples passed.  Returns true when any did ("used").
struct FiveSample { float alpha0, depth0, depth_max; bool pass0; };
__device__ __forceinline__ bool five_sample(const float4 A, const float4 B, float rpx, float rpy, float pixfx, float pixfy, float cT[5],
                                            FiveSample& o) {
  const float offx[5] = {0.0f, -0.5f, 0.5f, -0.5f, 0.5f}, offy[5] = {0.0f, -0.5f, -0.5f, 0.5f, 0.5f};
  bool used = false;
  o.pass0 = false;
  o.depth_max = -INFINITY;
#pragma unroll
  for (int c = 0; c < 5; c++) {
    const float dx = A.x - pixfx - offx[c], dy = A.y - pixfy - offy[c];
    const float depth = B.w + (rpx * dx + rpy * dy);
    const float power = -0.5f * (A.z * dx * dx + B.x * dy * dy) - A.w * dx * dy;
    if (power > 0.0f || power < B.z) continue;  // B.z: conservative exponent threshold for alpha < 1/255
    const float alpha = fminf(0.99f, B.y * exp_spec(power));
    if (alpha < 1.0f / 255.0f) continue;
    const float test_T = cT[c] * (1 - alpha);
    if (test_T < 0.0001f) continue;
    if (c == 0) { o.pass0 = true; o.alpha0 = alpha; o.depth0 = depth; }
    o.depth_max = fmaxf(o.depth_max, depth);
    cT[c] = test_T;
    used = true;
  }
  return used;
}

// Phase 2 walks the tile list again for every batch of 4 query points of a pixel and needs, per (pixel, entry), only WHETHER the
// 5-sample test of phase 1 let the entry through ("used": the reference keeps those ids in a 2048-entry per-thread array,
// forward.cu:1003,1121).  Round 1 replayed the test -- five specified exponentials per pair and pass; since round 4 phase 1 leaves one
// bit per (pixel, entry) in wave-private LDS (64 bits per lane and batch of 64 entries, 12 batches = 6 KB) and phase 2 reads it:
// the same decisions by construction, no exponential at all in phase 2.  Tiles with more than 768 entries replay as before.
constexpr int kUsedBatches = 12;   // 16: 5.26 ms on C2 with 4 M points, 12 / 10: 4.87 (LDS per wave decides the occupancy; + 3 KB of per-pixel staging for the point-major phase 2)

__global__ void __launch_bounds__(64) integrate_kernel(const IntegrateArgs a) {
  __shared__ float4 lds_a[64 * 4];
  __shared__ float4 lds_i[64 * 2];
  __shared__ unsigned long long lds_used[kUsedBatches * 64];
  const int item = xcd_band_remap(blockIdx.x, gridDim.x);
  const int tile = item >> 2, sub = item & 3;
  const int tile_x = tile % a.gx, tile_y = tile / a.gx;
  const int lane = threadIdx.x, lx = lane & 15, lr = lane >> 4;
  const int px = tile_x * 16 + lx, py = tile_y * 16 + sub * 4 + lr;
  const int W = a.W, H = a.H;
  const size_t HW = (size_t)H * W;
  const bool inside = px < W && py < H;
  const size_t pix = (size_t)py * W + px;
  const float pixfx = (float)px + 0.5f, pixfy = (float)py + 0.5f;
  // sample positions of this wave's strip (pixel centres +- 0.5) for the batch cull
  const float reg_x0 = (float)(tile_x * 16), reg_x1 = reg_x0 + 16.0f;
  const float reg_y0 = (float)(tile_y * 16 + sub * 4), reg_y1 = reg_y0 + 4.0f;
  const uint2 range = a.ranges[tile];
  const int n = (int)(range.y - range.x);

  // ---------------------------------------------------------------- phase 1: the image ----
  float cT[5] = {1.f, 1.f, 1.f, 1.f, 1.f};  // cT[0] is the pixel's T
  float C0 = 0.f, C1 = 0.f, C2 = 0.f, C3 = 0.f, C4 = 0.f, C6 = 0.f, C7 = 0.f;
  float mid_dc = 0.f, mid_px = 0.f, mid_py = 0.f, mid_mx = 0.f, mid_my = 0.f;
  uint32_t last_c = 0, n_local = 0;
  bool done = !inside;
  for (int base = 0; base < n; base += 64) {
    if (__all(done)) break;
    __syncthreads();
    const int k = base + lane;
    bool rel_lane = false;
    if (k < n) {
      const uint32_t g = a.point_list[range.x + k];
      const float4* src = a.splat_a + 4 * (size_t)g;
      const float4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
      lds_a[lane * 4 + 0] = q0; lds_a[lane * 4 + 1] = q1; lds_a[lane * 4 + 2] = q2; lds_a[lane * 4 + 3] = q3;
      rel_lane = entry_may_touch(q0, q1, reg_x0, reg_x1, reg_y0, reg_y1);
    }
    uint64_t rel = __ballot(rel_lane);
    __syncthreads();
    unsigned long long used_bits = 0ull;
    while (rel != 0) {
      const int j = __builtin_ctzll(rel);
      rel &= rel - 1;
      if (done) continue;
      const float4 A = lds_a[j * 4 + 0], B = lds_a[j * 4 + 1], Cc = lds_a[j * 4 + 2], D = lds_a[j * 4 + 3];
      const float T = cT[0];
      FiveSample f;
      if (!five_sample(A, B, Cc.w, D.x, pixfx, pixfy, cT, f)) continue;
      used_bits |= 1ull << j;
      if (f.pass0) {
        C0 += Cc.x * f.alpha0 * T; C1 += Cc.y * f.alpha0 * T; C2 += Cc.z * f.alpha0 * T;
      }
      if (f.depth_max > C6) C6 = f.depth_max;
      if (f.pass0) {
        C7 += f.alpha0 * T;
        C3 += f.depth0 * f.alpha0 * T;
        if (T > 0.5f) { C4 = f.depth0; mid_dc = B.w; mid_px = Cc.w; mid_py = D.x; mid_mx = A.x; mid_my = A.y; }
      }
      last_c = (uint32_t)(base + j + 1);
      n_local += 1;
      if (n_local >= (uint32_t)kMaxContributors) done = true;  // the reference stops this pixel here (forward.cu:1121-1125)
    }
    if ((base >> 6) < kUsedBatches) lds_used[(base >> 6) * 64 + lane] = used_bits;
  }
  const bool cached = n <= kUsedBatches * 64;   // wave-uniform: every batch of this tile has its bits (else phase 2 replays the test)
  const float T = cT[0];
  float col0 = 0.f, col1 = 0.f, col2 = 0.f;
  if (inside) {
    col0 = C0 + T * a.bg[0]; col1 = C1 + T * a.bg[1]; col2 = C2 + T * a.bg[2];
    a.final_T[pix] = T;
    a.n_contrib[pix] = last_c;
    a.out9[0 * HW + pix] = col0; a.out9[1 * HW + pix] = col1; a.out9[2 * HW + pix] = col2;
    a.out9[3 * HW + pix] = C3; a.out9[4 * HW + pix] = C4; a.out9[6 * HW + pix] = C6; a.out9[7 * HW + pix] = C7;
  }

  // --------------------------------------------------- phase 2: this pixel's query points ----
  uint32_t cur = 0, pe = 0;
  if (inside) {
    cur = pix == 0 ? 0u : a.pix_incl[pix - 1];
    pe = a.pix_incl[pix];
    a.out9[8 * HW + pix] = (float)(pe - cur);
  }
  if (cached) {
    // ---- point-major phase 2 (round 4): one query point per LANE.  The per-pixel form below gives every pixel-lane four point slots
    // per walk of the list; a pixel holds 1.6 points on average (C2, 4 M points) and an entry is used by a quarter of a strip's pixels,
    // so ~one lane-slot in ten does work.  Here the strip's points (sorted by pixel: four runs of pt_sorted) are dealt to the lanes 64 at
    // a time; a point-lane reads ITS pixel's used bits and per-pixel results from LDS and carries one point through the walk.
    __shared__ float lds_pix[64 * 8];      // per pixel of the strip: colour (3), median plane (depth, px, py, mx, my)
    __shared__ uint32_t lds_off[64 + 1];   // exclusive scan of the pixels' point counts
    __shared__ uint32_t lds_first[64];     // first index of the pixel's points in pt_sorted
    __shared__ uint32_t lds_last[64];      // the pixel's last contributor (bound of its walk)
    const uint32_t cnt = pe - cur;
    uint32_t incl = cnt;                   // inclusive wave scan of the counts
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t y = (uint32_t)__shfl_up((int)incl, d);
      if (lane >= d) incl += y;
    }
    const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
    __syncthreads();
    lds_off[lane] = incl - cnt;
    if (lane == 63) lds_off[64] = total;
    lds_first[lane] = cur;
    lds_last[lane] = last_c;
    lds_pix[lane * 8 + 0] = col0; lds_pix[lane * 8 + 1] = col1; lds_pix[lane * 8 + 2] = col2;
    lds_pix[lane * 8 + 3] = mid_dc; lds_pix[lane * 8 + 4] = mid_px; lds_pix[lane * 8 + 5] = mid_py;
    lds_pix[lane * 8 + 6] = mid_mx; lds_pix[lane * 8 + 7] = mid_my;
    __syncthreads();
    for (uint32_t r0 = 0; r0 < total; r0 += 64) {
      const uint32_t q = r0 + (uint32_t)lane;
      const bool have = q < total;
      // the pixel this point belongs to: the last p with off[p] <= q (binary search over the 64 offsets)
      int p = 0;
      if (have) {
#pragma unroll
        for (int step = 32; step > 0; step >>= 1)
          if (lds_off[p + step] <= q) p += step;
      }
      uint32_t pid1 = 0;
      float qx1 = 0.f, qy1 = 0.f, qd1 = 0.f, pa1 = 0.f, pT1 = 1.f;
      if (have) {
        pid1 = a.pt_sorted[lds_first[p] + (q - lds_off[p])];
        const float2 qq = a.p2d[pid1];
        qx1 = qq.x; qy1 = qq.y; qd1 = a.pdepth[pid1];
      }
      const uint32_t my_last = have ? lds_last[p] : 0u;
      for (int base = 0; base < n; base += 64) {
        if (__all((uint32_t)base >= my_last)) break;
        __syncthreads();
        const int k = base + lane;
        bool rel_lane = false;
        if (k < n) {
          const uint32_t g = a.point_list[range.x + k];
          const float4* src = a.splat_a + 4 * (size_t)g;
          const float4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
          lds_a[lane * 4 + 0] = q0; lds_a[lane * 4 + 1] = q1; lds_a[lane * 4 + 2] = q2; lds_a[lane * 4 + 3] = q3;
          const float4* si = a.inte_rec + 2 * (size_t)g;
          lds_i[lane * 2 + 0] = si[0]; lds_i[lane * 2 + 1] = si[1];
          rel_lane = entry_may_touch(q0, q1, reg_x0, reg_x1, reg_y0, reg_y1);
        }
        uint64_t rel = __ballot(rel_lane);
        __syncthreads();
        const unsigned long long my_bits = have ? lds_used[(base >> 6) * 64 + p] : 0ull;
        while (rel != 0) {
          const int j = __builtin_ctzll(rel);
          rel &= rel - 1;
          const bool mine = ((my_bits >> j) & 1ull) != 0ull;     // phase 1's decision for (this point's pixel, entry)
          if (!__any(mine)) continue;
          if (!mine) continue;
          const float4 A = lds_a[j * 4 + 0], B = lds_a[j * 4 + 1], Cc = lds_a[j * 4 + 2], D = lds_a[j * 4 + 3];
          const float4 I0 = lds_i[j * 2 + 0], I1 = lds_i[j * 2 + 1];
          const m3 inv = mk33(I0.x, I0.y, I0.z, I0.y, I0.w, I1.x, I0.z, I1.x, I1.y);
          const bool cond = I1.z != 0.0f;
          const float dx = A.x - qx1, dy = A.y - qy1;
          const float depth = B.w + (Cc.w * dx + D.x * dy);
          float alpha;
          if (cond) {
            const v3 du = mk3(dx, dy, B.w - fminf(qd1, depth));
            const float power = -0.5f * dot(du, mul(inv, du));
            alpha = fminf(0.99f, B.y * exp_spec(fminf(power, 80.0f)));
          } else if (qd1 < depth) {
            alpha = 0.f;
          } else {
            const v3 du = mk3(dx, dy, B.w);
            const float power = -0.5f * dot(du, mul(inv, du));
            alpha = fminf(0.99f, B.y * exp_spec(fminf(power, 80.0f)));
          }
          if (alpha < 1.0f / 255.0f) continue;
          const float test_T = pT1 * (1 - alpha);
          pa1 += alpha * pT1;
          pT1 = test_T;
        }
      }
      if (have) {
        const size_t qi = pid1;
        const float* pp = lds_pix + p * 8;
        a.out_alpha_integrated[qi] = pa1;
        a.out_color_integrated[3 * qi] = pp[0]; a.out_color_integrated[3 * qi + 1] = pp[1]; a.out_color_integrated[3 * qi + 2] = pp[2];
        a.out_coordinate2d[2 * qi] = qx1; a.out_coordinate2d[2 * qi + 1] = qy1;
        if (qd1 > 0) {
          const float dx = pp[6] - qx1, dy = pp[7] - qy1;
          const float depth = pp[3] + (pp[4] * dx + pp[5] * dy);
          a.out_sdf[qi] = depth - qd1;
        }
      }
    }
    return;
  }
  while (__any(cur < pe)) {
    const int np = (int)min((uint32_t)kPointsPerPass, pe - cur);
    uint32_t pid[kPointsPerPass];
    float qx[kPointsPerPass], qy[kPointsPerPass], qd[kPointsPerPass], pa[kPointsPerPass], pT[kPointsPerPass];
#pragma unroll
    for (int i = 0; i < kPointsPerPass; i++) {
      pid[i] = 0; qx[i] = qy[i] = qd[i] = 0.f; pa[i] = 0.f; pT[i] = 1.f;
      if (i < np) {
        pid[i] = a.pt_sorted[cur + i];
        const float2 q = a.p2d[pid[i]];
        qx[i] = q.x; qy[i] = q.y; qd[i] = a.pdepth[pid[i]];
      }
    }
    float rT[5] = {1.f, 1.f, 1.f, 1.f, 1.f};
    const uint32_t my_last = np > 0 ? last_c : 0u;
    for (int base = 0; base < n; base += 64) {
      if (__all((uint32_t)base >= my_last)) break;
      __syncthreads();
      const int k = base + lane;
      bool rel_lane = false;
      if (k < n) {
        const uint32_t g = a.point_list[range.x + k];
        const float4* src = a.splat_a + 4 * (size_t)g;
        const float4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
        lds_a[lane * 4 + 0] = q0; lds_a[lane * 4 + 1] = q1; lds_a[lane * 4 + 2] = q2; lds_a[lane * 4 + 3] = q3;
        const float4* si = a.inte_rec + 2 * (size_t)g;
        lds_i[lane * 2 + 0] = si[0]; lds_i[lane * 2 + 1] = si[1];
        rel_lane = entry_may_touch(q0, q1, reg_x0, reg_x1, reg_y0, reg_y1);
      }
      uint64_t rel = __ballot(rel_lane);
      __syncthreads();
      const unsigned long long my_bits = cached ? lds_used[(base >> 6) * 64 + lane] : 0ull;
      while (rel != 0) {
        const int j = __builtin_ctzll(rel);
        rel &= rel - 1;
        if (cached) {                                   // phase 1's decision for this (pixel, entry)
          const bool mine = np > 0 && ((my_bits >> j) & 1ull) != 0ull;
          if (!__any(mine)) continue;
          if (!mine) continue;
        } else if ((uint32_t)(base + j + 1) > my_last) continue;
        const float4 A = lds_a[j * 4 + 0], B = lds_a[j * 4 + 1], Cc = lds_a[j * 4 + 2], D = lds_a[j * 4 + 3];
        if (!cached) {
          FiveSample f;
          if (!five_sample(A, B, Cc.w, D.x, pixfx, pixfy, rT, f)) continue;
        }
        const float4 I0 = lds_i[j * 2 + 0], I1 = lds_i[j * 2 + 1];
        const m3 inv = mk33(I0.x, I0.y, I0.z, I0.y, I0.w, I1.x, I0.z, I1.x, I1.y);
        const bool cond = I1.z != 0.0f;
#pragma unroll
        for (int i = 0; i < kPointsPerPass; i++) {
          if (i >= np) continue;
          const float dx = A.x - qx[i], dy = A.y - qy[i];
          const float depth = B.w + (Cc.w * dx + D.x * dy);
          float alpha;
          if (cond) {
            const v3 du = mk3(dx, dy, B.w - fminf(qd[i], depth));
            const float power = -0.5f * dot(du, mul(inv, du));
            alpha = fminf(0.99f, B.y * exp_spec(fminf(power, 80.0f)));
          } else if (qd[i] < depth) {
            alpha = 0.f;
          } else {
            const v3 du = mk3(dx, dy, B.w);
            const float power = -0.5f * dot(du, mul(inv, du));
            alpha = fminf(0.99f, B.y * exp_spec(fminf(power, 80.0f)));
          }
          if (alpha < 1.0f / 255.0f) continue;
          const float test_T = pT[i] * (1 - alpha);
          pa[i] += alpha * pT[i];
          pT[i] = test_T;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < kPointsPerPass; i++) {
      if (i >= np) continue;
      const size_t q = pid[i];
      a.out_alpha_integrated[q] = pa[i];
      a.out_color_integrated[3 * q] = col0; a.out_color_integrated[3 * q + 1] = col1; a.out_color_integrated[3 * q + 2] = col2;
      a.out_coordinate2d[2 * q] = qx[i]; a.out_coordinate2d[2 * q + 1] = qy[i];
      if (qd[i] > 0) {
        const float dx = mid_mx - qx[i], dy = mid_my - qy[i];
        const float depth = mid_dc + (mid_px * dx + mid_py * dy);
        a.out_sdf[q] = depth - qd[i];
      }
    }
    cur += (uint32_t)np;
  }
}

// =========================================================================== blend, fwd ==
struct BlendFwdArgs {
  const uint2* ranges; const uint32_t* point_list; const float4* splat_a; const float4* splat_b;
  int W, H, gx, ntiles; float focal_x, focal_y;
  const float* bg;
  float* out_color; float* out_coord; float* out_mcoord; float* out_depth; float* out_mdepth; float* out_alpha; float* out_normal;
  uint32_t* n_contrib; float* accum_coord; float* accum_depth; float* normal_length;
  const uint32_t* blk_count; const uint32_t* blk_base; uint32_t* blk_consumed; uint32_t* blk_chunks; const uint32_t* blk_order;   // sub-tile entry streams (rg_streams.inc)
};

// The maps a mode does NOT produce are all-zero in the reference, whatever the flags (torch::full(0), rasterize_points.cu:71-77).  A
// caller that hands their pointers over gets them zeroed here, by the pixel's own lane in the forward's epilogue -- the stores ride along
// in a VALU-bound kernel; a separate fill of the 6 unproduced planes of a depth-mode 1080p view was a 9-us kernel + its launch gap per
// forward.  (NULL: the caller does not want them, or provides zeros itself.)
template <bool COORD, bool DEPTH>
__device__ __forceinline__ void zero_unproduced_maps(const BlendFwdArgs& a, size_t pix, size_t HW) {
  // non-temporal: nothing on this path reads these planes again, and 50 MB of ordinary stores would push the entry streams and records the
  // backward is about to re-read out of the L2 / Infinity Cache
  auto z = [](float* p) { __builtin_nontemporal_store(0.0f, p); };
  if constexpr (!COORD) {
    if (a.out_coord) { z(a.out_coord + pix); z(a.out_coord + HW + pix); z(a.out_coord + 2 * HW + pix); }
    if (a.out_mcoord) { z(a.out_mcoord + pix); z(a.out_mcoord + HW + pix); z(a.out_mcoord + 2 * HW + pix); }
  }
  if constexpr (!DEPTH) {
    if (a.out_depth) z(a.out_depth + pix);
    if (a.out_mdepth) z(a.out_mdepth + pix);
  }
  if constexpr (!COORD && !DEPTH) {
    if (a.out_normal) { z(a.out_normal + pix); z(a.out_normal + HW + pix); z(a.out_normal + 2 * HW + pix); }
  }
}

// blockIdx -> work item such that each XCD (block b runs on XCD b % 8) owns a contiguous band.
__device__ __forceinline__ int xcd_band_remap(int b, int n) {
  const int q = n >> 3, r = n & 7, xcd = b & 7, loc = b >> 3;
  return xcd * q + (xcd < r ? xcd : r) + loc;
}

// One wave64 owns a 16 x (4*PPL) strip of the tile: lane -> column (lane & 15), rows (lane >> 4) + 4 s.
struct StripGeom { int px, py_first; float rx0, rx1, ry0, ry1; };
template <int PPL>
__device__ __forceinline__ StripGeom lane_geometry(int lane, int tile_x, int tile_y, int sub) {
  StripGeom g;
  g.px = tile_x * 16 + (lane & 15); g.py_first = tile_y * 16 + sub * (4 * PPL) + (lane >> 4);
  g.rx0 = (float)(tile_x * 16); g.rx1 = g.rx0 + 15.0f; g.ry0 = (float)(tile_y * 16 + sub * (4 * PPL)); g.ry1 = g.ry0 + (float)(4 * PPL - 1);
  return g;
}
constexpr int kStripRowStep = 4;   // rows between the pixels of one lane

template <bool COORD, bool DEPTH, int PPL>
__global__ void __launch_bounds__(64) blend_fwd_kernel(const BlendFwdArgs a) {
  constexpr bool NORMAL = COORD || DEPTH;
  constexpr int WPT = 4 / PPL;  // waves per tile
  __shared__ float4 lds_a[65 * 4];
  __shared__ float4 lds_b[COORD ? 64 * 3 : 1];

  const int item = xcd_band_remap(blockIdx.x, gridDim.x);
  const int tile = item / WPT, sub = item - tile * WPT;
  const int tile_x = tile % a.gx, tile_y = tile / a.gx;
  const int lane = threadIdx.x;
  const StripGeom geo = lane_geometry<PPL>(lane, tile_x, tile_y, sub);
  const int px = geo.px;
  const int py0 = geo.py_first;  // slot s -> row py0 + 4 s
  const int W = a.W, H = a.H;
  const size_t HW = (size_t)H * W;
  const float pixfx = (float)px;
  // pixel rectangle owned by this wave (for batch culling)
  const float reg_x0 = geo.rx0, reg_x1 = geo.rx1, reg_y0 = geo.ry0, reg_y1 = geo.ry1;

  const uint2 range = a.ranges[tile];
  const int n = (int)(range.y - range.x);

  // Tw is the working transmittance: it equals T until the pixel terminates, then it is forced to
  // 0 -- any later candidate then fails `T*(1-alpha) >= 1e-4` by itself, which is exactly "done".
  float pixfy[PPL], T[PPL], Tw[PPL], Cr[PPL], Cg[PPL], Cb[PPL], weight[PPL];
  float Dep[PPL], mDep[PPL], Nx[PPL], Ny[PPL], Nz[PPL];
  float Co[COORD ? PPL : 1][3], mCo[COORD ? PPL : 1][3];
  uint32_t last_c[PPL], max_c[PPL];
  bool inside[PPL];
#pragma unroll
  for (int s = 0; s < PPL; s++) {
    const int py = py0 + kStripRowStep * s;
    pixfy[s] = (float)py;
    inside[s] = px < W && py < H;
    T[s] = 1.0f; Tw[s] = inside[s] ? 1.0f : 0.0f; Cr[s] = Cg[s] = Cb[s] = 0.f; weight[s] = 0.f;
    Dep[s] = mDep[s] = 0.f; Nx[s] = Ny[s] = Nz[s] = 0.f;
    last_c[s] = 0; max_c[s] = 0xFFFFFFFFu;
    if constexpr (COORD) {
#pragma unroll
      for (int c = 0; c < 3; c++) { Co[s][c] = 0.f; mCo[s][c] = 0.f; }
    }
  }
  bool all_done;
  {
    bool d = true;
#pragma unroll
    for (int s = 0; s < PPL; s++) d = d && (Tw[s] == 0.0f);
    all_done = __all(d);
  }

  for (int base = 0; base < n && !all_done; base += 64) {
    // ---- stage up to 64 list entries: one 64-B record (one cache line) per lane ----
    __syncthreads();
    const int k = base + lane;
    bool rel_lane = false;
    if (k < n) {
      const uint32_t g = a.point_list[range.x + k];
      const float4* src = a.splat_a + 4 * (size_t)g;
      const float4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
      lds_a[lane * 4 + 0] = q0; lds_a[lane * 4 + 1] = q1; lds_a[lane * 4 + 2] = q2; lds_a[lane * 4 + 3] = q3;
      if constexpr (COORD) {
        const float4* sb = a.splat_b + 3 * (size_t)g;
        lds_b[lane * 3 + 0] = sb[0]; lds_b[lane * 3 + 1] = sb[1]; lds_b[lane * 3 + 2] = sb[2];
      }
      rel_lane = entry_may_touch(q0, q1, reg_x0, reg_x1, reg_y0, reg_y1);
    }
    uint64_t rel = __ballot(rel_lane);
    const int niter = (int)__popcll(rel);
    __syncthreads();
    for (int it = 0; it < niter && !all_done; it++) {   // scalar trip count
      const int j = __builtin_ctzll(rel);
      rel &= rel - 1;
      const float4 A = lds_a[j * 4 + 0], B = lds_a[j * 4 + 1];  // {mx,my,cx,cy} {cz,op,thr,ts}
      const float dx = A.x - pixfx;
      const float a_x = (A.z * dx) * dx;
      const float b_xy = A.w * dx;
      float power[PPL];
      bool cand[PPL], anyc = false;
#pragma unroll
      for (int s = 0; s < PPL; s++) {
        const float dy = A.y - pixfy[s];
        power[s] = splat_power(a_x, b_xy, B.x, dy);
        cand[s] = !(power[s] > 0.0f) && !(power[s] < B.z);
        anyc = anyc || cand[s];
      }
      if (__any(anyc)) {  // (no `continue`: a single loop back-edge keeps the per-pixel state in place, no PHI copies)
      const float4 C = lds_a[j * 4 + 2], Dq = lds_a[j * 4 + 3];  // {r,g,b,rpx} {rpy,nx,ny,nz}
      float4 E0, E1, E2;
      if constexpr (COORD) { E0 = lds_b[j * 3 + 0]; E1 = lds_b[j * 3 + 1]; E2 = lds_b[j * 3 + 2]; }
      const uint32_t contributor = (uint32_t)(base + j + 1);
      bool newly_done = false;
#pragma unroll
      for (int s = 0; s < PPL; s++) {
        if (cand[s]) {
          const float G = exp_spec(power[s]);
          const float alpha = fminf(0.99f, B.y * G);
          if (!(alpha < 1.0f / 255.0f)) {
            const float test_T = Tw[s] * (1 - alpha);
            if (test_T < 0.0001f) {
              newly_done = newly_done || (Tw[s] != 0.0f);
              Tw[s] = 0.0f;
            } else {
              const float aT = alpha * T[s];
              const float dy = A.y - pixfy[s];
              Cr[s] = fmaf(C.x, aT, Cr[s]); Cg[s] = fmaf(C.y, aT, Cg[s]); Cb[s] = fmaf(C.z, aT, Cb[s]);
              const bool before_median = T[s] > 0.5f;
              if constexpr (COORD) {
                const float c0 = fmaf(E0.y, dy, fmaf(E0.x, dx, E1.z));
                const float c1 = fmaf(E0.w, dy, fmaf(E0.z, dx, E1.w));
                const float c2 = fmaf(E1.y, dy, fmaf(E1.x, dx, E2.x));
                Co[s][0] = fmaf(c0, aT, Co[s][0]); Co[s][1] = fmaf(c1, aT, Co[s][1]); Co[s][2] = fmaf(c2, aT, Co[s][2]);
                if (before_median) { mCo[s][0] = c0; mCo[s][1] = c1; mCo[s][2] = c2; }
              }
              if constexpr (DEPTH) {
                const float t = B.w + fmaf(C.w, dx, Dq.x * dy);
                Dep[s] = fmaf(t, aT, Dep[s]);
                if (before_median) mDep[s] = t;
              }
              if constexpr (NORMAL) {
                Nx[s] = fmaf(Dq.y, aT, Nx[s]); Ny[s] = fmaf(Dq.z, aT, Ny[s]); Nz[s] = fmaf(Dq.w, aT, Nz[s]);
                if (before_median) max_c[s] = contributor;
              }
              weight[s] += aT;
              T[s] = test_T;
              Tw[s] = test_T;
              last_c[s] = contributor;
            }
          }
        }
      }
      if (__any(newly_done)) {
        bool d = true;
#pragma unroll
        for (int s = 0; s < PPL; s++) d = d && (Tw[s] == 0.0f);
        all_done = __all(d);
      }
      }
    }
  }

  // ---- epilogue (forward.cu:631-692) ----
  const float pnx = (pixfx - W / 2.f) / a.focal_x;
#pragma unroll
  for (int s = 0; s < PPL; s++) {
    if (!inside[s]) continue;
    const size_t pix = (size_t)W * (py0 + kStripRowStep * s) + px;
    const float pny = (pixfy[s] - H / 2.f) / a.focal_y;
    const float ln = sqrtf(pnx * pnx + pny * pny + 1);
    a.n_contrib[pix] = last_c[s];
    a.n_contrib[pix + HW] = max_c[s];
    a.out_color[pix] = fmaf(T[s], a.bg[0], Cr[s]);
    a.out_color[HW + pix] = fmaf(T[s], a.bg[1], Cg[s]);
    a.out_color[2 * HW + pix] = fmaf(T[s], a.bg[2], Cb[s]);
    a.out_alpha[pix] = weight[s];
    zero_unproduced_maps<COORD, DEPTH>(a, pix, HW);
    if constexpr (COORD) {
#pragma unroll
      for (int c = 0; c < 3; c++) {
        a.out_coord[c * HW + pix] = last_c[s] ? Co[s][c] / weight[s] : 0.f;
        a.accum_coord[c * HW + pix] = Co[s][c];
        a.out_mcoord[c * HW + pix] = mCo[s][c];
      }
    }
    if constexpr (DEPTH) {
      const float depth_ln = Dep[s] / ln;
      a.accum_depth[pix] = depth_ln;
      a.out_depth[pix] = last_c[s] ? depth_ln / weight[s] : 0.f;
      a.out_mdepth[pix] = mDep[s] / ln;
    }
    if constexpr (NORMAL) {
      if (last_c[s]) {
        float len_n = sqrtf(Nx[s] * Nx[s] + Ny[s] * Ny[s] + Nz[s] * Nz[s]);
        a.normal_length[pix] = len_n;
        len_n = fmaxf(len_n, 1.0E-12F);
        a.out_normal[pix] = Nx[s] / len_n;
        a.out_normal[HW + pix] = Ny[s] / len_n;
        a.out_normal[2 * HW + pix] = Nz[s] / len_n;
      } else {
        a.normal_length[pix] = 1;
        a.out_normal[pix] = 0; a.out_normal[HW + pix] = 0; a.out_normal[2 * HW + pix] = 0;
      }
    }
  }
}

// =========================================================================== blend, bwd ==
// 1/x for x in [0.01, 1]: hardware reciprocal (1 ulp) + one Newton step.  T is recovered back to front as T <- T / (1 - alpha)
// over hundreds of entries (backward.cu:843), so the per-step error compounds; the reference divides exactly (no fast-math in
// its build).  With the refinement the chain is as accurate as an IEEE division at 3 instructions instead of ~10.
__device__ __forceinline__ float rcp_refined(float x) {
  const float r = __builtin_amdgcn_rcpf(x);
  return fmaf(fmaf(-x, r, 1.0f), r, r);
}

struct BlendBwdArgs {
  const uint2* ranges; const uint32_t* point_list; const float4* splat_a; const float4* splat_b;
  int W, H, gx, ntiles; float focal_x, focal_y;
  const float* bg;
  const float* alphas; const float* normalmap;
  const uint32_t* n_contrib; const float* accum_coord; const float* accum_depth; const float* normal_length;
  const float* dL_dpix; const float* dL_dcoord; const float* dL_dmcoord; const float* dL_ddepth; const float* dL_dmdepth;
  const float* dL_dalpha; const float* dL_dnormal;
  float* acc;  // [P][REC] per-Gaussian sums, SplatAcc order
  int P;       // Gaussians (rows of acc)
  const uint32_t* stream_tag;   // ImageState::stream_tag (stream kernels only)
  uint32_t* stream_err;         // mapped host word: set when stream_tag says this buffer holds no entry streams (may be nullptr)
  const uint32_t* blk_base; const uint32_t* blk_consumed; const uint32_t* blk_chunks; const uint32_t* blk_order;   // sub-tile entry streams (rg_streams.inc)
};

// In: v[i] = this lane's partial sum of component i.  Out (return value): the wave-wide total of
// component (lane & (N-1)).  Each butterfly stage halves the live components while doubling the
// lanes summed: lanes whose stage bit is set keep the upper half of the components, the others the
// lower half, and every lane hands the half it drops to a partner of the opposite class.  The four
// in-row stages use DPP (no LDS crossbar): row_ror:8 (= xor 8), row_half_mirror (bit 2 flips,
// bit 3 kept), quad_perm xor 2, quad_perm xor 1 -- together they span all 16 lanes of a row.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float x) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(x), CTRL, 0xF, 0xF, false));
}
template <int HALF, int BIT, int CTRL>
__device__ __forceinline__ void bfly_stage_dpp(float* v, int lane) {
  const bool up = (lane >> BIT) & 1;
#pragma unroll
  for (int i = 0; i < HALF; i++) {
    const float send = up ? v[i] : v[i + HALF];
    const float keep = up ? v[i + HALF] : v[i];
    v[i] = keep + dpp_mov<CTRL>(send);
  }
}
template <int HALF, int BIT>
__device__ __forceinline__ void bfly_stage_xor(float* v, int lane) {   // a stage that crosses the DPP rows (ds_bpermute)
  const bool up = (lane >> BIT) & 1;
#pragma unroll
  for (int i = 0; i < HALF; i++) {
    const float send = up ? v[i] : v[i + HALF];
    const float keep = up ? v[i + HALF] : v[i];
    v[i] = keep + __shfl_xor(send, 1 << BIT);
  }
}
template <int N>
__device__ __forceinline__ float wave_reduce_scatter(float (&v)[N], int lane) {
  if constexpr (N == 32) bfly_stage_xor<16, 4>(v, lane);
  bfly_stage_dpp<8, 3, 0x128>(v, lane);  // row_ror:8
  bfly_stage_dpp<4, 2, 0x141>(v, lane);  // row_half_mirror
  bfly_stage_dpp<2, 1, 0x4E>(v, lane);   // quad_perm [2,3,0,1]
  bfly_stage_dpp<1, 0, 0xB1>(v, lane);   // quad_perm [1,0,3,2]
  float r = v[0];
  if constexpr (N == 16) r += __shfl_xor(r, 16);
  r += __shfl_xor(r, 32);
  return r;
}

typedef float v4f __attribute__((ext_vector_type(4)));

// ---- per-(block, entry) reduction of the 16 per-Gaussian sums over the 16 lanes of a DPP row (round 6) ----
// Rounds 2-5 ran all four butterfly stages on the VALU (24 bank-masked DPP adds + 13 selects / adds = 37 half-rate instructions,
// ~165 of the loop's ~530 issue cycles).  Now only the first stage does: a DPP add with a bank mask only writes the lanes of the
// enabled banks (4 lanes each), so "lanes 0..7 keep components 0..7, lanes 8..15 keep 8..15" is two masked adds per component
// (row_ror:8 pairs lane l with l ^ 8).  The remaining 8 x 8 transposition goes through wave-private LDS: every lane stores its 8
// partial sums (two ds_write_b128), lane l reads component l & 7 of the 8 lanes of its half row (four ds_read2_b32) and adds them
// up with 7 full-rate adds: 16 DPP + 7 adds on the VALU (~90 cycles), 6 LDS instructions that other waves' VALU work covers.
// (All sixteen components through LDS would need 4 KB per wave -- with the staged records 8.7 KB, four waves per SIMD.)
// Layout of a row's 576-byte scratch (144 floats): lane j's 8 sums at float offset (j >> 3) * 72 + (j & 7) * 8, its two 16-byte
// halves swapped when bit 2 of j is set; the +32 bytes per half row and +64 per row put the two half rows and the two rows that share
// an LDS cycle on different banks for the reads, the swap does the same for the eight lanes of a ds_write_b128 group.
constexpr int kRedRowFloats = 144;
__device__ __forceinline__ void row_reduce16_first_stage(float (&v)[16]) {
  asm volatile(
      "s_nop 1\n\t"
      "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %0, %8, %8 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %1, %9, %9 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %2, %10, %10 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %3, %11, %11 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %4, %12, %12 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %5, %5, %5 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %5, %13, %13 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %6, %14, %14 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "v_add_f32_dpp %7, %7, %7 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
      "v_add_f32_dpp %7, %15, %15 row_ror:8 row_mask:0xf bank_mask:0xc\n\t"
      "s_nop 1"
      : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
      : "v"(v[8]), "v"(v[9]), "v"(v[10]), "v"(v[11]), "v"(v[12]), "v"(v[13]), "v"(v[14]), "v"(v[15]));
}
struct RowReduceAddr {   // per lane, set up once per kernel
  float* w0; float* w1;         // where this lane's two 16-byte halves go
  const float* r0; const float* r1;   // component (lane & 7) of source lanes 0..3 / 4..7 of the half row (+ 8 k floats for source k)
};
__device__ __forceinline__ RowReduceAddr row_reduce_addr(float* scratch, int grp, int l) {
  RowReduceAddr a;
  float* row = scratch + grp * kRedRowFloats;
  const int sw = (l >> 2) & 1, c8 = l & 7, half = l >> 3;
  a.w0 = row + half * 72 + c8 * 8 + sw * 4;
  a.w1 = row + half * 72 + c8 * 8 + (sw ^ 1) * 4;
  a.r0 = row + half * 72 + (c8 >> 2) * 4 + (c8 & 3);
  a.r1 = row + half * 72 + ((c8 >> 2) ^ 1) * 4 + (c8 & 3);
  return a;
}
// In: v[c] = this lane's partial sum of component c.  Returns the row total of component (lane & 15).
__device__ __forceinline__ float row_reduce16(float (&v)[16], const RowReduceAddr& ad) {
  row_reduce16_first_stage(v);
  *reinterpret_cast<v4f*>(ad.w0) = v4f{v[0], v[1], v[2], v[3]};
  *reinterpret_cast<v4f*>(ad.w1) = v4f{v[4], v[5], v[6], v[7]};
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  const float t0 = ad.r0[0], t1 = ad.r0[8], t2 = ad.r0[16], t3 = ad.r0[24];
  const float t4 = ad.r1[32], t5 = ad.r1[40], t6 = ad.r1[48], t7 = ad.r1[56];
  __builtin_amdgcn_wave_barrier();   // the next call's stores stay behind these loads
  return ((t0 + t1) + (t2 + t3)) + ((t4 + t5) + (t6 + t7));
}

// ------------------------------------------------------------------ blend, bwd (packed) ----
// Second formulation of the same backward: the pixels of one lane are handled in PAIRS as 2-wide
// fp32 vectors (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 do two pixels per issue slot), and a
// pixel that does not take part in an entry is switched off with two selects (alpha := 0, G := 0)
// instead of a divergent branch: with alpha = 0 every recurrence below is an exact no-op
// (T*rcp(1) = T, acc + 0*d = acc) and every gradient term is 0.  The only branches left are
// wave-uniform, so the pair bodies are straight-line code the scheduler can interleave.
typedef float f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f2 fma2(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 bc2(float v) { return f2{v, v}; }

// second launch-bound = waves per SIMD the register allocator must leave room for
template <bool COORD, bool DEPTH, int PPL>
__global__ void __launch_bounds__(64, (COORD ? (PPL == 4 ? 1 : 2) : (PPL == 4 ? 2 : 5))) blend_bwd_packed_kernel(const BlendBwdArgs a) {
  static_assert(PPL == 2 || PPL == 4, "pairs of pixels per lane");
  constexpr bool NORMAL = COORD || DEPTH;
  constexpr int NP = PPL / 2;  // pairs per lane
  constexpr int WPT = 4 / PPL;
  constexpr int REC = COORD ? 32 : 16;
  __shared__ float4 lds_a[65 * 4];
  __shared__ float4 lds_b[COORD ? 64 * 3 : 1];
  __shared__ uint32_t lds_id[65];
  __shared__ __attribute__((aligned(16))) float lds_red[4 * kRedRowFloats];   // row_reduce16's scratch (rg_streams.inc)

  const int item = xcd_band_remap(blockIdx.x, gridDim.x);
  const int tile = item / WPT, sub = item - tile * WPT;
  const int tile_x = tile % a.gx, tile_y = tile / a.gx;
  const int lane = threadIdx.x;
  const StripGeom geo = lane_geometry<PPL>(lane, tile_x, tile_y, sub);
  const int px = geo.px;
  const int py0 = geo.py_first;
  const int W = a.W, H = a.H;
  const size_t HW = (size_t)H * W;
  const float pixfx = (float)px;
  const float reg_x0 = geo.rx0, reg_x1 = geo.rx1, reg_y0 = geo.ry0, reg_y1 = geo.ry1;
  const uint2 range = a.ranges[tile];

  // Q is the ONE "behind" accumulator per pixel.  Upstream keeps one per blended quantity
  // (accum_rec[3], accum_t_rec, accum_normal_rec[3], accum_alpha_rec, accum_coord_rec[3]:
  // backward.cu:870,900,930,949,962), each following  acc <- acc + alpha*(v - acc)  and each entering
  // dL/dalpha as  w*(v - acc)  with a per-pixel constant weight w (the pixel's cotangent).  The recurrence is
  // linear, so Q = sum_k w_k*acc_k obeys  Q <- Q + alpha*(V - Q)  with  V = sum_k w_k*v_k,  and
  // sum_k w_k*(v_k - acc_k) = V - Q: identical mathematics, 1 register and 2 operations instead of 8 and 24.
  f2 pixfy[NP], T[NP], Q[NP], dLa[NP], tb[NP];
  f2 dLc[NP][3];
  f2 dLt[NP], dLmt[NP];
  f2 dLn[NP][3];
  f2 dLco[COORD ? NP : 1][3], dLmco[COORD ? NP : 1][3];
  uint32_t last_c[PPL], max_cm1[PPL];
  uint32_t wave_last = 0;
  const float pnx = (pixfx - W / 2.f) / a.focal_x;
#pragma unroll
  for (int s = 0; s < PPL; s++) {
    const int q = s >> 1, e = s & 1;
    const int py = py0 + kStripRowStep * s;
    pixfy[q][e] = (float)py;
    const bool inside = px < W && py < H;
    const size_t pix = inside ? (size_t)W * py + px : 0;
    const float alpha_px = inside ? a.alphas[pix] : 0.f;
    const float T_final = inside ? (1 - alpha_px) : 0.f;
    const float w_final = alpha_px;
    T[q][e] = T_final;
    last_c[s] = inside ? a.n_contrib[pix] : 0u;
    max_cm1[s] = (inside ? a.n_contrib[pix + HW] : 0u) - 1u;
    wave_last = max(wave_last, last_c[s]);
    Q[q][e] = 0.f;
    dLt[q][e] = 0.f; dLmt[q][e] = 0.f;
    float dl3[3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
      dl3[c] = inside ? a.dL_dpix[c * HW + pix] : 0.f;
      dLc[q][c][e] = dl3[c];
      dLn[q][c][e] = 0.f;
      if constexpr (COORD) { dLco[q][c][e] = 0.f; dLmco[q][c][e] = 0.f; }
    }
    float dla = inside ? a.dL_dalpha[pix] : 0.f;
    tb[q][e] = -T_final * (a.bg[0] * dl3[0] + a.bg[1] * dl3[1] + a.bg[2] * dl3[2]);
    // Pixels nothing blended into (alpha = 0) cannot pass a gradient to any Gaussian; their 1/alpha factors would be inf/NaN and
    // poison the wave-wide sums through the multiplicative masks, so their geometry cotangents stay zero.
    if (NORMAL && inside && last_c[s] > 0) {
      const float ww = w_final * w_final;
      const float pny = ((float)py - H / 2.f) / a.focal_y;
      const float ln = sqrtf(pnx * pnx + pny * pny + 1);
      if constexpr (COORD) {
#pragma unroll
        for (int c = 0; c < 3; c++) {
          const float gw = a.dL_dcoord[c * HW + pix];
          dla -= gw * a.accum_coord[c * HW + pix] / ww;
          dLco[q][c][e] = gw / w_final;
          dLmco[q][c][e] = a.dL_dmcoord[c * HW + pix];
        }
      }
      if constexpr (DEPTH) {
        const float gw = a.dL_ddepth[pix];
        dla -= gw * a.accum_depth[pix] / ww;
        dLt[q][e] = gw / w_final / ln;
        dLmt[q][e] = a.dL_dmdepth[pix] / ln;
      }
      {
        const float g0 = a.dL_dnormal[pix], g1 = a.dL_dnormal[HW + pix], g2 = a.dL_dnormal[2 * HW + pix];
        const float n0 = a.normalmap[pix], n1 = a.normalmap[HW + pix], n2 = a.normalmap[2 * HW + pix];
        const float nlen = a.normal_length[pix];
        if (nlen < 1.0E-12F) {
          dLn[q][0][e] = g0 / 1.0E-12F; dLn[q][1][e] = g1 / 1.0E-12F; dLn[q][2][e] = g2 / 1.0E-12F;
        } else {
          const float dt = g0 * n0 + g1 * n1 + g2 * n2;
          dLn[q][0][e] = (g0 - dt * n0) / nlen; dLn[q][1][e] = (g1 - dt * n1) / nlen; dLn[q][2][e] = (g2 - dt * n2) / nlen;
        }
      }
    }
    dLa[q][e] = dla;
  }
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) wave_last = max(wave_last, (uint32_t)__shfl_xor((int)wave_last, m));
  const f2 cW = bc2(0.5f * W), cH = bc2(0.5f * H);
  const RowReduceAddr red = row_reduce_addr(lds_red, lane >> 4, lane & 15);

  for (int hi = (int)wave_last; hi > 0; hi -= 64) {
    __syncthreads();
    const int e0 = hi - 1 - lane;
    bool rel_lane = false;
    if (e0 >= 0) {
      const uint32_t g = a.point_list[range.x + e0];
      lds_id[lane] = g;
      const float4* src = a.splat_a + 4 * (size_t)g;
      const float4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
      lds_a[lane * 4 + 0] = q0; lds_a[lane * 4 + 1] = q1; lds_a[lane * 4 + 2] = q2; lds_a[lane * 4 + 3] = q3;
      if constexpr (COORD) {
        const float4* sb = a.splat_b + 3 * (size_t)g;
        lds_b[lane * 3 + 0] = sb[0]; lds_b[lane * 3 + 1] = sb[1]; lds_b[lane * 3 + 2] = sb[2];
      }
      rel_lane = entry_may_touch(q0, q1, reg_x0, reg_x1, reg_y0, reg_y1);
    }
    uint64_t rel = __ballot(rel_lane);
    const int niter = (int)__popcll(rel);
    __syncthreads();
    for (int it = 0; it < niter; it++) {   // scalar trip count
      const int j = __builtin_ctzll(rel);  // LDS slot j holds list position hi-1-j: ascending j = back to front
      rel &= rel - 1;
      const float4 A = lds_a[j * 4 + 0], B = lds_a[j * 4 + 1], C = lds_a[j * 4 + 2], Dq = lds_a[j * 4 + 3];
      const uint32_t gid = lds_id[j];
      const uint32_t pos = (uint32_t)(hi - 1 - j);
      const float dx = A.x - pixfx;
      const float a_x = (A.z * dx) * dx;
      const float b_xy = A.w * dx;
      f2 dy[NP], power[NP];
      bool cand[PPL], anyc = false;
#pragma unroll
      for (int q = 0; q < NP; q++) {
        dy[q] = bc2(A.y) - pixfy[q];
        const f2 sq = bc2(a_x) + (bc2(B.x) * dy[q]) * dy[q];
        const f2 vq = bc2(b_xy) * dy[q];
        power[q] = fma2(bc2(-0.5f), sq, -vq);  // == splat_power(), one rounding (rg_blend.h)
#pragma unroll
        for (int e = 0; e < 2; e++) {
          const float pw = power[q][e];
          cand[2 * q + e] = (pos < last_c[2 * q + e]) && !(pw > 0.0f) && !(pw < B.z);
          anyc = anyc || cand[2 * q + e];
        }
      }
      if (!__any(anyc)) continue;
      float4 E0, E1, E2;
      if constexpr (COORD) { E0 = lds_b[j * 3 + 0]; E1 = lds_b[j * 3 + 1]; E2 = lds_b[j * 3 + 2]; }
      // the lane's sums over its pixel pairs; components 9..14 as RAW MOMENTS of h = opacity G dL/dalpha about the Gaussian's centre, everything
      // "times dx" applied once to the lane's total (all pixels of a lane share dx): the record blend_bwd_streams_kernel writes
      // (rg_streams.inc), turned into the reference's sums once per Gaussian by preprocess_bwd_kernel (PreBwdArgs::acc_raw)
      f2 s_col[3], s_nrm[3], s_dt = bc2(0.f), s_dty = bc2(0.f), s_u = bc2(0.f), s_h = bc2(0.f), s_uy = bc2(0.f), s_uyy = bc2(0.f), s_ab = bc2(0.f);
      f2 s_co[COORD ? 3 : 1], s_coy[COORD ? 3 : 1];
#pragma unroll
      for (int c = 0; c < 3; c++) { s_col[c] = bc2(0.f); s_nrm[c] = bc2(0.f); }
      if constexpr (COORD) {
#pragma unroll
        for (int c = 0; c < 3; c++) { s_co[c] = bc2(0.f); s_coy[c] = bc2(0.f); }
      }
      bool contributed = false;
      const float dxcx = dx * A.z;
#pragma unroll
      for (int q = 0; q < NP; q++) {
        if (!__any(cand[2 * q] || cand[2 * q + 1])) continue;  // wave-uniform
        // ---- alpha (decision = forward's exp_spec rule; value from the hardware exp) ----
        f2 G = f2{__expf(power[q][0]), __expf(power[q][1])};
        f2 a_raw = bc2(B.y) * G;
        {
          const bool b0 = fabsf(fmaf(a_raw[0], 255.0f, -1.0f)) < 1.0e-4f, b1 = fabsf(fmaf(a_raw[1], 255.0f, -1.0f)) < 1.0e-4f;
          if (__any(b0 || b1)) {  // within 1e-4 of the 1/255 threshold: decide with the specified exponential
            if (b0) { G[0] = exp_spec(power[q][0]); a_raw[0] = B.y * G[0]; }
            if (b1) { G[1] = exp_spec(power[q][1]); a_raw[1] = B.y * G[1]; }
          }
        }
        f2 alpha = f2{fminf(0.99f, a_raw[0]), fminf(0.99f, a_raw[1])};
        const bool act0 = cand[2 * q] && !(alpha[0] < 1.0f / 255.0f), act1 = cand[2 * q + 1] && !(alpha[1] < 1.0f / 255.0f);
        contributed = contributed || act0 || act1;
        alpha = f2{act0 ? alpha[0] : 0.f, act1 ? alpha[1] : 0.f};
        G = f2{act0 ? G[0] : 0.f, act1 ? G[1] : 0.f};
        const f2 one_m_a = bc2(1.f) - alpha;
        const f2 inv1ma = f2{rcp_refined(one_m_a[0]), rcp_refined(one_m_a[1])};
        T[q] = T[q] * inv1ma;
        const f2 dch = alpha * T[q];
        // V = <cotangent of this pixel, blended quantities of this Gaussian>; dL/dalpha's blend part = V - Q
        f2 V = dLa[q];
        {
          const float col[3] = {C.x, C.y, C.z};
#pragma unroll
          for (int c = 0; c < 3; c++) {
            V = fma2(bc2(col[c]), dLc[q][c], V);
            s_col[c] = fma2(dch, dLc[q][c], s_col[c]);
          }
        }
        const bool med0 = act0 && pos == max_cm1[2 * q], med1 = act1 && pos == max_cm1[2 * q + 1];
        if constexpr (COORD) {
          const float cpx[3] = {E0.x, E0.z, E1.x}, cpy[3] = {E0.y, E0.w, E1.y}, vp[3] = {E1.z, E1.w, E2.x};
#pragma unroll
          for (int c = 0; c < 3; c++) {
            const f2 cc = fma2(bc2(cpy[c]), dy[q], bc2(fmaf(cpx[c], dx, vp[c])));
            V = fma2(cc, dLco[q][c], V);
            const f2 msel = f2{med0 ? dLmco[q][c][0] : 0.f, med1 ? dLmco[q][c][1] : 0.f};
            const f2 dco = fma2(dch, dLco[q][c], msel);
            s_co[c] += dco;
            s_coy[c] = fma2(dco, dy[q], s_coy[c]);
          }
        }
        if constexpr (DEPTH) {
          const f2 t = fma2(bc2(Dq.x), dy[q], bc2(fmaf(C.w, dx, B.w)));
          V = fma2(t, dLt[q], V);
          const f2 msel = f2{med0 ? dLmt[q][0] : 0.f, med1 ? dLmt[q][1] : 0.f};
          const f2 dt_ = fma2(dch, dLt[q], msel);
          s_dt += dt_;
          s_dty = fma2(dt_, dy[q], s_dty);
        }
        if constexpr (NORMAL) {
          const float nn[3] = {Dq.y, Dq.z, Dq.w};
#pragma unroll
          for (int c = 0; c < 3; c++) {
            V = fma2(bc2(nn[c]), dLn[q][c], V);
            s_nrm[c] = fma2(dch, dLn[q][c], s_nrm[c]);
          }
        }
        f2 dL_dopa = V - Q[q];
        Q[q] = fma2(alpha, dL_dopa, Q[q]);   // alpha = 0 for a pixel that sits this entry out: Q unchanged
        dL_dopa = dL_dopa * T[q];
        dL_dopa = fma2(inv1ma, tb[q], dL_dopa);

        const f2 u = G * dL_dopa;
        const f2 hq = bc2(B.y) * u;   // h = opacity * u: the moments are h's (rg_streams.inc)
        const f2 uy = hq * dy[q];
        const f2 ex = fma2(dy[q], bc2(A.w), bc2(dxcx));    // the conic applied to (dx, dy)
        const f2 ey = fma2(dy[q], bc2(B.x), bc2(b_xy));
        const f2 tt = fma2(__builtin_elementwise_abs(ey), cH, __builtin_elementwise_abs(ex) * cW);
        s_u += u; s_h += hq; s_uy += uy;
        s_uyy = fma2(uy, dy[q], s_uyy);
        s_ab = fma2(__builtin_elementwise_abs(hq), tt, s_ab);
      }
      const uint64_t contrib_mask = __ballot(contributed);
      if (contrib_mask == 0) continue;
      float gs[REC];
      gs[0] = s_col[0][0] + s_col[0][1]; gs[1] = s_col[1][0] + s_col[1][1]; gs[2] = s_col[2][0] + s_col[2][1];
      gs[3] = s_dt[0] + s_dt[1]; gs[4] = gs[3] * dx; gs[5] = s_dty[0] + s_dty[1];
      gs[6] = s_nrm[0][0] + s_nrm[0][1]; gs[7] = s_nrm[1][0] + s_nrm[1][1]; gs[8] = s_nrm[2][0] + s_nrm[2][1];
      gs[15] = s_u[0] + s_u[1];
      gs[9] = (s_h[0] + s_h[1]) * dx; gs[10] = s_uy[0] + s_uy[1]; gs[11] = s_ab[0] + s_ab[1];
      gs[12] = gs[9] * dx; gs[13] = gs[10] * dx; gs[14] = s_uyy[0] + s_uyy[1];
      if constexpr (COORD) {
#pragma unroll
        for (int c = 0; c < 3; c++) { gs[16 + c] = s_co[c][0] + s_co[c][1]; gs[19 + 2 * c] = gs[16 + c] * dx; gs[20 + 2 * c] = s_coy[c][0] + s_coy[c][1]; }
#pragma unroll
        for (int c = 25; c < 32; c++) gs[c] = 0.f;
      }
      // rows first (one DPP stage + wave-private LDS: row_reduce16), then the four rows' totals of component (lane & 15)
      float tot = row_reduce16(*reinterpret_cast<float (*)[16]>(gs), red);
      tot += __shfl_xor(tot, 16);
      tot += __shfl_xor(tot, 32);
      if constexpr (REC == 32) {
        float tot1 = row_reduce16(*reinterpret_cast<float (*)[16]>(gs + 16), red);
        tot1 += __shfl_xor(tot1, 16);
        tot1 += __shfl_xor(tot1, 32);
        if (lane >= 16 && lane < 25) unsafeAtomicAdd(a.acc + (size_t)gid * REC + lane, tot1);
      }
      if (lane < 16) unsafeAtomicAdd(a.acc + (size_t)gid * REC + lane, tot);
    }
  }
}

// ======================================================================= preprocess, bwd ==
struct PreBwdArgs {
  int P, D, M;
  const float* means3D; const float* scales; const float* rotations; const float* cov3D_precomp; const float* shs;
  const int* radii; const float4* splat_a; const uint8_t* clamped; const float* acc; int rec;
  CamArgs cam;
  float* dL_dmean2D; float* dL_dcolor; float* dL_dopacity; float* dL_dmean3D; float* dL_dcov3D; float* dL_dsh; float* dL_dscale;
  float* dL_drot;
  float* dL_drgb_clamped;  // optional [P,3]: dL/dRGB with the clamp mask applied (the view-parallel factored exchange)
  int opacity_grad_intended;  // RadegsBwdArgs::opacity_grad_intended (include/radegs.h)
  int drgb_done;              // dL_drgb_clamped was already written by drgb_clamped_kernel (RadegsBwdArgs::drgb_ready)
  int acc_final;              // the records hold the reference's FINAL per-Gaussian sums (constant factors applied): radegs_backward_from_sums
  int acc_raw;                // components 9..14 of the records are raw moments of u = G dL/dalpha (blend_bwd_streams_kernel, rg_streams.inc):
                              // the mean2D / conic sums are formed here, once per Gaussian; 2: and the record is written back in the
                              // reference's form (debugging / tests: RadegsBwdArgs::keep_sums)
  float* acc_out;             // acc_raw == 2: where the converted record goes (the accumulator itself)
  int acc_rezero;             // clear every consumed record (RadegsBwdArgs::acc_reuse): the accumulator goes back to its owner all zeros
  int vec_slab;               // the SH slab moves in 16-byte pieces (3M % 4 == 0, 3M <= 48, shs and dL_dsh 16-byte aligned)
  int first_block;            // this launch covers the Gaussians from first_block * 128 on (RadegsBwdArgs::grad_chunks)
};

// dL/dRGB with the SH clamp mask applied, straight from the blend backward's sums (the first three floats of every accumulator
// record): what the factored view-parallel exchange all-gathers.  Its own kernel so that the collective can start one kernel
// earlier, under preprocess_bwd_kernel (RadegsBwdArgs::drgb_ready).
__global__ void __launch_bounds__(256) drgb_clamped_kernel(int P, const int* __restrict__ radii, const uint8_t* __restrict__ clamped,
                                                           const float* __restrict__ acc, int rec, float* __restrict__ out) {
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= P) return;
  float r = 0.f, g = 0.f, b = 0.f;
  if (radii[idx] > 0) {
    const unsigned cl = (unsigned)clamped[idx];
    const float* a = acc + (size_t)idx * rec;
    r = a[0] * ((cl & 1u) ? 0.f : 1.f); g = a[1] * ((cl & 2u) ? 0.f : 1.f); b = a[2] * ((cl & 4u) ? 0.f : 1.f);
  }
  out[3 * (size_t)idx] = r; out[3 * (size_t)idx + 1] = g; out[3 * (size_t)idx + 2] = b;
}

// 128 Gaussians per block.  The (P,M,3) SH tensor and its gradient are 192-byte rows at SH degree 3: read or
// written by one thread each they would be 64 different cache lines per instruction.  The block therefore
// moves its contiguous 128-row slab with coalesced accesses through LDS (row stride 3M+1 words: odd, so the
// per-thread row walks are bank-conflict free); sh and dL/dsh share the slab (sh_bwd's access order allows it).
constexpr int kPreBwdThreads = 128;   // 64 / 256 measured in round 4: no difference
// Copies a block's contiguous [nrows][rowf] slab between global memory and the LDS slab of row stride rowf + 1, 128 consecutive
// words per step.  (row, column) of word e come from a multiply-high by the reciprocal of the run-time row length (exact for
// the slab's few thousand words): a true division per word cost more than everything else the kernel does, and carrying
// (row, column) from step to step serialises the loads.
template <bool TO_LDS>
__device__ __forceinline__ void slab_copy(float* slab, float* gmem, int nrows, int rowf, int tid) {
  const int stride = rowf + 1, n = nrows * rowf;
  const uint32_t magic = 0xFFFFFFFFu / (uint32_t)rowf + 1u;   // ceil(2^32 / rowf): floor(e / rowf) == mulhi(e, magic) for e * rowf < 2^32
#pragma unroll 4
  for (int e = tid; e < n; e += kPreBwdThreads) {
    const int g = (int)__umulhi((uint32_t)e, magic), c = e - g * rowf;
    if constexpr (TO_LDS) slab[g * stride + c] = gmem[e];
    else gmem[e] = slab[g * stride + c];
  }
}

// Memory-level parallelism (round 4).  The kernel moves ~670 B per Gaussian and computes for ~5 000 instructions at 3 waves per SIMD:
// what it cannot afford is a chain of dependent memory latencies.  The first version copied the slab with 4-byte loads in a loop the
// compiler unrolled by 8 -- 2 KB in flight per wave, six full latencies per block one after the other -- and only then, behind the
// barrier and the visibility test, asked for the Gaussian's own records: ~6 MB in flight on the whole chip, which at ~1.5 us of loaded
// latency is the 3.5 TB/s it ran at.  Now EVERY global read of a block is issued before anything waits: the slab as 12 x 16 bytes per
// thread (rows of 3M floats with 3M % 4 == 0 and 16-byte aligned tensors, i.e. SH degree 3 and 1; other shapes keep the word loop),
// the accumulator record, mean, scale, rotation and flags of the thread's Gaussian (for invisible ones too: the record is there and
// reading it costs less than waiting for `radii` first).  LDS side: row stride 3M + 1 words, so a 16-byte piece goes in as four words;
// neighbouring lanes are 4 words apart and rows shift by one word, which keeps the 64 lanes of a store on different banks.
constexpr int kSlabVecs = 12;   // 16-byte pieces per thread: 128 rows x 48 floats / 128 threads
__global__ void __launch_bounds__(kPreBwdThreads) preprocess_bwd_kernel(const PreBwdArgs a) {
  extern __shared__ float sh_slab[];  // [128][3M+1]
  const int tid = threadIdx.x;
  const int base = ((int)blockIdx.x + a.first_block) * kPreBwdThreads;
  const int idx = base + tid;
  const int nrows = min(kPreBwdThreads, a.P - base);
  const int rowf = a.M * 3, stride = rowf + 1;
  const bool have_sh = a.shs != nullptr;
  const bool vec = have_sh && a.vec_slab != 0;   // host: rowf % 4 == 0, rowf <= 4 * kSlabVecs, shs and dL_dsh 16-byte aligned
  const int rowf4 = rowf >> 2, n4 = nrows * rowf4;
  const uint32_t magic4 = vec ? 0xFFFFFFFFu / (uint32_t)rowf4 + 1u : 0u;   // floor(e / rowf4) == mulhi(e, magic4), as in slab_copy

  // ---- every global read of the block ----
  float4 v[kSlabVecs];
  if (vec) {
    const float4* g4 = reinterpret_cast<const float4*>(a.shs + (size_t)base * rowf);
#pragma unroll
    for (int k = 0; k < kSlabVecs; k++) {
      const int e4 = tid + k * kPreBwdThreads;
      v[k] = e4 < n4 ? g4[e4] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  const bool live = idx < a.P;
  const size_t i = live ? (size_t)idx : 0;
  const bool has_sr = a.scales != nullptr;
  int radius = 0;
  unsigned cflags = 0;
  float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0, r3 = r0, r4 = r0, r5 = r0, rq = r0;
  float r6 = 0.f, m0 = 0.f, m1 = 0.f, m2 = 0.f, s0 = 0.f, s1 = 0.f, s2 = 0.f;
  if (live) {
    radius = a.radii[idx];
    const float4* r = reinterpret_cast<const float4*>(a.acc + i * a.rec);
    r0 = r[0]; r1 = r[1]; r2 = r[2]; r3 = r[3];
    if (a.rec == 32) { r4 = r[4]; r5 = r[5]; r6 = r[6].x; }
    if (a.acc_rezero && radius > 0) {   // only a visible Gaussian's record can have been touched (it is in no list otherwise)
      float4* w = reinterpret_cast<float4*>(a.acc_out + i * a.rec);
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      w[0] = z; w[1] = z; w[2] = z; w[3] = z;
      if (a.rec == 32) { w[4] = z; w[5] = z; w[6] = z; }
    }
    m0 = a.means3D[3 * i]; m1 = a.means3D[3 * i + 1]; m2 = a.means3D[3 * i + 2];
    if (has_sr) {
      s0 = a.scales[3 * i]; s1 = a.scales[3 * i + 1]; s2 = a.scales[3 * i + 2];
      rq = *reinterpret_cast<const float4*>(a.rotations + 4 * i);
    }
    cflags = (unsigned)a.clamped[idx];   // bits 0..2: SH clamp flags (written for visible Gaussians only; unused otherwise)
  }

  if (vec) {
#pragma unroll
    for (int k = 0; k < kSlabVecs; k++) {
      const int e4 = tid + k * kPreBwdThreads;
      if (e4 < n4) {
        const int g = (int)__umulhi((uint32_t)e4, magic4), c = (e4 - g * rowf4) << 2;
        float* d = sh_slab + g * stride + c;
        d[0] = v[k].x; d[1] = v[k].y; d[2] = v[k].z; d[3] = v[k].w;
      }
    }
  } else if (have_sh) {
    slab_copy<true>(sh_slab, const_cast<float*>(a.shs) + (size_t)base * rowf, nrows, rowf, tid);
  }
  if (have_sh) __syncthreads();

  if (live) {
    float* row = have_sh ? sh_slab + tid * stride : nullptr;
    if (!(radius > 0)) {  // invisible: every returned row is zero (rasterize_points.cu:180-193)
#pragma unroll
      for (int c = 0; c < 3; c++) { a.dL_dmean2D[3 * i + c] = 0; a.dL_dcolor[3 * i + c] = 0; a.dL_dmean3D[3 * i + c] = 0; a.dL_dscale[3 * i + c] = 0; }
      a.dL_dopacity[i] = 0;
#pragma unroll
      for (int c = 0; c < 6; c++) a.dL_dcov3D[6 * i + c] = 0;
#pragma unroll
      for (int c = 0; c < 4; c++) a.dL_drot[4 * i + c] = 0;
      if (row) for (int c = 0; c < rowf; c++) row[c] = 0;
      if (a.dL_drgb_clamped && !a.drgb_done) { a.dL_drgb_clamped[3 * i] = 0; a.dL_drgb_clamped[3 * i + 1] = 0; a.dL_drgb_clamped[3 * i + 2] = 0; }
    } else {
      const Camera cam = load_camera(a.cam);
      SplatAcc acc;
      acc.dcolor[0] = r0.x; acc.dcolor[1] = r0.y; acc.dcolor[2] = r0.z; acc.dts = r0.w;
      acc.drp[0] = r1.x; acc.drp[1] = r1.y; acc.dnrm[0] = r1.z; acc.dnrm[1] = r1.w;
      acc.dnrm[2] = r2.x; acc.dmean2D[0] = r2.y; acc.dmean2D[1] = r2.z; acc.dmean2D[2] = r2.w;
      acc.dconic[0] = r3.x; acc.dconic[1] = r3.y; acc.dconic[2] = r3.z; acc.dop = r3.w;
      if (a.rec == 32) {
        acc.dvp[0] = r4.x; acc.dvp[1] = r4.y; acc.dvp[2] = r4.z; acc.dcp[0] = r4.w;
        acc.dcp[1] = r5.x; acc.dcp[2] = r5.y; acc.dcp[3] = r5.z; acc.dcp[4] = r5.w; acc.dcp[5] = r6;
      } else {
        acc.dvp[0] = acc.dvp[1] = acc.dvp[2] = 0.f;
#pragma unroll
        for (int c = 0; c < 6; c++) acc.dcp[c] = 0.f;
      }
      acc.raw = a.acc_raw != 0;   // components 9..14 are raw moments (rg_streams.inc): preprocess_bwd() turns them into the reference's sums
      // constant factors the blend backward left out of its sums (linear, so they commute with the sum):
      // 1/focal on the plane gradients (backward.cu:917-922,939-940), W/2 and H/2 on mean2D (:1002-1003)
      if (!a.acc_final) {
        const float ifx = 1.0f / cam.focal_x, ify = 1.0f / cam.focal_y;
        acc.drp[0] *= ifx; acc.drp[1] *= ify;
#pragma unroll
        for (int c = 0; c < 3; c++) { acc.dcp[2 * c] *= ifx; acc.dcp[2 * c + 1] *= ify; }
        acc.half_wh = true;   // W/2, H/2 on mean2D are applied by preprocess_bwd() (after a raw record's conversion)
      }
      float sc3[3] = {s0, s1, s2}, rq4[4] = {rq.x, rq.y, rq.z, rq.w};
      float cov[6];
      if (a.cov3D_precomp) {
#pragma unroll
        for (int c = 0; c < 6; c++) cov[c] = a.cov3D_precomp[6 * i + c];
      } else {
        cov3d_from_scale_rot(sc3, cam.scale_modifier, rq4, cov);
      }
      // what the reference's computeCov2DCUDA reads as `conic_opacity[idx].w` is dL_dconic[idx].w (argument slip at
      // rasterizer_impl.cu:568); the stored opacity*coef only with opacity_grad_intended (include/radegs.h)
      // (a raw record holds sum h dy dy there: dL_dconic.w = -1/2 of it, rg_streams.inc)
      const float op_combined = a.opacity_grad_intended ? a.splat_a[4 * i + 1].y : (a.acc_raw ? -0.5f * acc.dconic[2] : acc.dconic[2]);
      if (row) {  // rows beyond the active degree stay zero
        const int K = (a.D + 1) * (a.D + 1);
        for (int c = K * 3; c < rowf; c++) row[c] = 0;
      }
      SplatBwd o;
      o.dscale[0] = o.dscale[1] = o.dscale[2] = 0; o.drot[0] = o.drot[1] = o.drot[2] = o.drot[3] = 0;
      preprocess_bwd(mk3(m0, m1, m2), has_sr ? sc3 : nullptr, has_sr ? rq4 : nullptr, cov, op_combined, a.D, row,
                     cflags & 7u, cam, acc, row, o);
      if (a.acc_raw == 2) {   // RadegsBwdArgs::keep_sums: the record goes back in the reference's form (before the W/2, H/2 factors)
        float4* w = reinterpret_cast<float4*>(a.acc_out + i * a.rec);
        w[2] = make_float4(r2.x, o.sums_mean2D[0], o.sums_mean2D[1], o.sums_mean2D[2]);
        w[3] = make_float4(o.sums_conic[0], o.sums_conic[1], o.sums_conic[2], acc.dop);
      }
#pragma unroll
      for (int c = 0; c < 3; c++) {
        a.dL_dmean2D[3 * i + c] = o.dmean2D[c];
        a.dL_dcolor[3 * i + c] = acc.dcolor[c];
        a.dL_dmean3D[3 * i + c] = o.dmean3D[c];
        a.dL_dscale[3 * i + c] = o.dscale[c];
      }
      a.dL_dopacity[i] = o.dopacity;
      if (a.dL_drgb_clamped && !a.drgb_done) {
        const unsigned cl = cflags & 7u;
#pragma unroll
        for (int c = 0; c < 3; c++) a.dL_drgb_clamped[3 * i + c] = acc.dcolor[c] * (((cl >> c) & 1u) ? 0.f : 1.f);
      }
#pragma unroll
      for (int c = 0; c < 6; c++) a.dL_dcov3D[6 * i + c] = o.dcov3D[c];
      *reinterpret_cast<float4*>(a.dL_drot + 4 * i) = make_float4(o.drot[0], o.drot[1], o.drot[2], o.drot[3]);
    }
  }
  if (have_sh && a.dL_dsh) {
    __syncthreads();
    if (vec) {
      float4* g4 = reinterpret_cast<float4*>(a.dL_dsh + (size_t)base * rowf);
#pragma unroll
      for (int k = 0; k < kSlabVecs; k++) {
        const int e4 = tid + k * kPreBwdThreads;
        if (e4 < n4) {
          const int g = (int)__umulhi((uint32_t)e4, magic4), c = (e4 - g * rowf4) << 2;
          const float* d = sh_slab + g * stride + c;
          g4[e4] = make_float4(d[0], d[1], d[2], d[3]);   // (non-temporal stores for these rows: no difference, same-box A/B 1.278-1.283 | 1.277-1.280 ms per step)
        }
      }
    } else {
      slab_copy<false>(sh_slab, a.dL_dsh + (size_t)base * rowf, nrows, rowf, tid);
    }
  }
}

// dL/dsh[P,M,3] = scale * sum over views v of  w(dir_v) (x) dRGB_v   -- rebuilds the SH gradient of a view-parallel batch
// from what the ranks all-gathered (12 B per Gaussian per view instead of all-reducing 192 B per Gaussian).
// Same 128-row LDS slab as preprocess_bwd_kernel for the coalesced write-out.
__global__ void __launch_bounds__(kPreBwdThreads) sh_grad_from_views_kernel(int P, int D, int M, int nviews, const float* __restrict__ means3D,
                                                                           const float* __restrict__ campos, const float* __restrict__ drgb,
                                                                           float scale, float* __restrict__ dL_dsh) {
  extern __shared__ float sh_slab[];
  const int tid = threadIdx.x, base = blockIdx.x * kPreBwdThreads, idx = base + tid;
  const int nrows = min(kPreBwdThreads, P - base);
  const int rowf = M * 3, stride = rowf + 1;
  if (idx < P) {
    float acc[48];
#pragma unroll
    for (int k = 0; k < 48; k++) acc[k] = 0.f;
    const v3 pos = mk3(means3D[3 * (size_t)idx], means3D[3 * (size_t)idx + 1], means3D[3 * (size_t)idx + 2]);
    for (int v = 0; v < nviews; v++) {
      const float* g = drgb + ((size_t)v * P + idx) * 3;
      const float gr = g[0], gg = g[1], gb = g[2];
      if (gr == 0.f && gg == 0.f && gb == 0.f) continue;  // not visible (or fully clamped) in this view
      const float cp[3] = {campos[3 * v], campos[3 * v + 1], campos[3 * v + 2]};
      float w[16];
      sh_basis(D, pos, cp, w);
#pragma unroll
      for (int k = 0; k < 16; k++) { acc[3 * k] += w[k] * gr; acc[3 * k + 1] += w[k] * gg; acc[3 * k + 2] += w[k] * gb; }
    }
    float* row = sh_slab + tid * stride;
    const int K3 = 3 * (D + 1) * (D + 1);
#pragma unroll
    for (int c = 0; c < 48; c++)
      if (c < rowf) row[c] = c < K3 ? acc[c] * scale : 0.f;
    for (int c = 48; c < rowf; c++) row[c] = 0.f;
  }
  __syncthreads();
  slab_copy<false>(sh_slab, dL_dsh + (size_t)base * rowf, nrows, rowf, tid);
}

}  // namespace rg

// The blend kernels over sub-tile entry streams use the helpers above.
#include "rg_streams.inc"

// The host-side orchestration and the C ABI (include/radegs.h) follow; they need the kernels above in scope.
#include "rg_launch.inc"
