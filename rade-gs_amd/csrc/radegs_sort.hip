// radegs_sort.hip -- hand-written stable LSD radix sort of (u32 key, u32 value) pairs for the binning stage.
//
// Replaces the two device-library sorts of round 1 (the counterpart of cub::DeviceRadixSort::SortPairs at
// DGR/cuda_rasterizer/rasterizer_impl.cu:376-381).  Both sorts on this path are small by HBM standards (1M
// depth keys; ~4M (tile, gaussian) instances with a 13-bit key), so the design goal is few, short kernels:
//
//   per 8-bit pass:   digit_histogram_kernel   one 256-bin histogram per 2048-item block   (reads keys)
//                     scan_rows_kernel         one workgroup per digit scans its row of block counts
//                     scatter_kernel           stable scatter                                (reads keys+values, writes both)
//
// Stability (what makes ties keep ascending Gaussian index) comes from the scatter's ranking: a wave
// owns a CONTIGUOUS run of the block's items and walks it 64 items at a time; within one step, lanes with equal
// digits find each other with 8 ballots (one per digit bit) and rank themselves by popcount of the lower lanes;
// across steps a per-wave LDS counter array carries the running count per digit; across waves the block first
// histograms every wave's run (LDS atomics) and prefix-sums those per digit.  No atomics on global memory, no
// inter-workgroup communication, deterministic output.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "rg_prims.h"

namespace rg {

namespace {

constexpr int kSortThreads = 256;                 // 4 waves
constexpr int kBins = 256;

// K: key type -- uint32_t (depth keys, tile ids of grids above 65 536 tiles) or uint16_t (tile ids: 13 bits at 1080p, 15 at 4K; a third
// less traffic per pass -- histogram 2 B, scatter 6 B in + 6 B out per item instead of 4 / 8 / 8).
// BINS: 256 (digits of up to 8 bits) or 512 (9-bit digits: the 27-bit depth sort in three passes).  key_base: subtracted from every key
// before its digit is taken (the depth sort's keys are float bits above bits(0.2f); stored keys stay as they are).
template <int ITEMS, class K, int BINS>
__global__ void __launch_bounds__(kSortThreads) digit_histogram_kernel(const K* __restrict__ keys, uint32_t n, int shift,
                                                                       uint32_t mask, uint32_t* __restrict__ hist, uint32_t nblocks,
                                                                       const uint32_t* __restrict__ n_dev, uint32_t key_base) {
  __shared__ uint32_t h[BINS];
  if (n_dev) n = min(n, *n_dev);  // capacity launch: the real count is still on the device (rg_launch.inc, speculative binning)
#pragma unroll
  for (int d = threadIdx.x; d < BINS; d += kSortThreads) h[d] = 0;
  __syncthreads();
  const uint32_t base = blockIdx.x * (uint32_t)(kSortThreads * ITEMS);
  uint32_t k[ITEMS];   // every load is issued before the first one is used (a load per LDS atomic would serialise their latencies)
#pragma unroll
  for (int r = 0; r < ITEMS; r++) {
    const uint32_t i = base + r * kSortThreads + threadIdx.x;
    k[r] = i < n ? (uint32_t)keys[i] : 0u;
  }
#pragma unroll
  for (int r = 0; r < ITEMS; r++) {
    const uint32_t i = base + r * kSortThreads + threadIdx.x;
    if (i < n) atomicAdd(&h[((k[r] - key_base) >> shift) & mask], 1u);
  }
  __syncthreads();
#pragma unroll
  for (int d = threadIdx.x; d < BINS; d += kSortThreads) hist[(size_t)d * nblocks + blockIdx.x] = h[d];  // [digit][block]
}

// One workgroup per digit: exclusive scan of that digit's per-block counts (in place) + the digit's total.
__global__ void __launch_bounds__(kSortThreads) scan_rows_kernel(uint32_t* __restrict__ hist, uint32_t nblocks, uint32_t* __restrict__ totals) {
  __shared__ uint32_t wave_sum[4];
  __shared__ uint32_t carry_s;
  uint32_t* row = hist + (size_t)blockIdx.x * nblocks;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (threadIdx.x == 0) carry_s = 0;
  __syncthreads();
  for (uint32_t start = 0; start < nblocks; start += kSortThreads) {
    const uint32_t i = start + threadIdx.x;
    const uint32_t v = i < nblocks ? row[i] : 0u;
    uint32_t x = v;  // inclusive scan inside the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t y = __shfl_up(x, d);
      if (lane >= d) x += y;
    }
    if (lane == 63) wave_sum[wave] = x;
    __syncthreads();
    uint32_t off = carry_s;
    for (int w = 0; w < wave; w++) off += wave_sum[w];
    if (i < nblocks) row[i] = off + x - v;
    __syncthreads();
    if (threadIdx.x == kSortThreads - 1) carry_s = off + x;
    __syncthreads();
  }
  if (threadIdx.x == 0) totals[blockIdx.x] = carry_s;
}

// exclusive scan of one value per thread across the 256-thread block
__device__ __forceinline__ uint32_t block256_exclusive(uint32_t t, uint32_t* tmp4) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t x = t;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = __shfl_up(x, d);
    if (lane >= d) x += y;
  }
  __syncthreads();
  if (lane == 63) tmp4[wave] = x;
  __syncthreads();
  uint32_t off = 0;
  for (int w = 0; w < wave; w++) off += tmp4[w];
  return off + x - t;
}

template <int ITEMS, class K, int BINS>
__global__ void __launch_bounds__(kSortThreads) scatter_kernel(const K* __restrict__ keys_in, const uint32_t* __restrict__ vals_in,
                                                               K* __restrict__ keys_out, uint32_t* __restrict__ vals_out, uint32_t n,
                                                               int shift, uint32_t mask, int nbits, const uint32_t* __restrict__ hist,
                                                               uint32_t nblocks, const uint32_t* __restrict__ totals,
                                                               const uint32_t* __restrict__ n_dev, uint32_t key_base) {
  constexpr int BLOCK_ITEMS = kSortThreads * ITEMS, WAVE_ITEMS = 64 * ITEMS;
  constexpr int PER = BINS / kSortThreads;    // consecutive digits per thread in the table work (1 or 2)
  if (n_dev) n = min(n, *n_dev);
  __shared__ uint32_t digit_base[BINS];       // global offset of this block's first item of each digit
  __shared__ uint32_t local_start[BINS];      // position of each digit's first item in the block-local sorted order
  // 16-bit counters (a wave's run has at most 64 * ITEMS = 2 048 items, a block 8 192): with ITEMS = 32 and 16-bit keys the workgroup's LDS
  // is 53.3 KB instead of 55.3 -- THREE workgroups per CU (160 KB) instead of two (round 6; C5's 50 M-instance tile sort)
  __shared__ uint16_t wave_cnt[4][BINS];      // histogram of each wave's run, then running rank counters
  __shared__ uint32_t scan_tmp[4];
  __shared__ K lds_k[BLOCK_ITEMS];            // the block's items reordered by digit (stable), so that the global
  __shared__ uint32_t lds_v[BLOCK_ITEMS];     // writes below go out in contiguous per-digit runs
  // ITEMS = 32 (tens of millions of items, C5): 2 x 32 KB + 6 KB of counters = 70 KB of LDS per workgroup -- above the 64 KB of every
  // AMD architecture before gfx950 (160 KB per CU).  This library is built for gfx950 only (build.py: ARCH).
  static_assert(2 * BLOCK_ITEMS * sizeof(uint32_t) + 8 * BINS * sizeof(uint32_t) <= 160 * 1024, "scatter_kernel: LDS footprint exceeds gfx950's 160 KB");
  static_assert(64 * ITEMS < 65536 && kSortThreads * ITEMS < 65536, "scatter_kernel: 16-bit counters");
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  {   // exclusive scan of the digit totals (thread t owns digits PER t .. PER t + PER - 1) + this block's offset inside each digit
    uint32_t tot[PER], sum = 0;
#pragma unroll
    for (int q = 0; q < PER; q++) { tot[q] = totals[PER * tid + q]; sum += tot[q]; }
    uint32_t ex = block256_exclusive(sum, scan_tmp);
#pragma unroll
    for (int q = 0; q < PER; q++) {
      digit_base[PER * tid + q] = ex + hist[(size_t)(PER * tid + q) * nblocks + blockIdx.x];
      ex += tot[q];
    }
  }
#pragma unroll
  for (int w = 0; w < 4; w++)
#pragma unroll
    for (int q = 0; q < PER; q++) wave_cnt[w][PER * tid + q] = 0;
  __syncthreads();
  const uint32_t block0 = blockIdx.x * (uint32_t)BLOCK_ITEMS;
  const uint32_t run0 = block0 + wave * (uint32_t)WAVE_ITEMS;  // this wave's contiguous run
  // the wave's whole run goes to registers first: all 2 x ITEMS loads are in flight together, the keys serve both phases
  uint32_t rk[ITEMS], rv[ITEMS];
#pragma unroll
  for (int r = 0; r < ITEMS; r++) {
    const uint32_t i = run0 + r * 64 + lane;
    rk[r] = i < n ? (uint32_t)keys_in[i] : 0xFFFFFFFFu;
  }
#pragma unroll
  for (int r = 0; r < ITEMS; r++) {
    const uint32_t i = run0 + r * 64 + lane;
    rv[r] = i < n ? (vals_in ? vals_in[i] : i) : 0u;
  }
  // ---- phase A: histogram of every wave's run ----
#pragma unroll
  for (int r = 0; r < ITEMS; r++) {
    const uint32_t i = run0 + r * 64 + lane;
    if (i < n) {   // (LDS atomics are 32-bit: the increment goes to the digit's half of its word; a half never carries -- counts <= 2 048)
      const uint32_t d = ((rk[r] - key_base) >> shift) & mask;
      atomicAdd(reinterpret_cast<uint32_t*>(&wave_cnt[wave][d & ~1u]), 1u << (16u * (d & 1u)));
    }
  }
  __syncthreads();
  // per digit: the 4 wave counts become exclusive prefixes (wave w starts after waves < w); block total per digit
  {
    uint32_t bc[PER], sum = 0;
#pragma unroll
    for (int q = 0; q < PER; q++) {
      uint32_t block_count = 0;
#pragma unroll
      for (int w = 0; w < 4; w++) {
        const uint32_t c = wave_cnt[w][PER * tid + q];
        wave_cnt[w][PER * tid + q] = (uint16_t)block_count;
        block_count += c;
      }
      bc[q] = block_count;
      sum += block_count;
    }
    uint32_t ex = block256_exclusive(sum, scan_tmp);
#pragma unroll
    for (int q = 0; q < PER; q++) { local_start[PER * tid + q] = ex; ex += bc[q]; }
  }
  __syncthreads();
  // ---- phase B: stable ranks, 64 consecutive items per step; items land in LDS in digit order ----
#pragma unroll
  for (int r = 0; r < ITEMS; r++) {
    const uint32_t i = run0 + r * 64 + lane;
    const bool valid = i < n;
    const uint32_t key = rk[r];
    const uint32_t val = rv[r];
    const uint32_t digit = ((key - key_base) >> shift) & mask;
    // lanes holding the same digit (invalid lanes only match each other and are never counted)
    uint64_t peers = __ballot(valid);
    if (!valid) peers = ~peers;
    for (int b = 0; b < nbits; b++) {
      const bool bit = (digit >> b) & 1u;
      const uint64_t m = __ballot(bit);
      peers &= bit ? m : ~m;
    }
    const uint64_t lower = peers & ((1ull << lane) - 1ull);
    const uint32_t rank_in_step = (uint32_t)__popcll(lower);
    const uint32_t count_in_step = (uint32_t)__popcll(peers);
    uint32_t before = 0;
    if (valid) before = wave_cnt[wave][digit];          // items of this digit earlier in the block order
    // the first lane of every peer group advances the running counter (one writer per digit: no atomic needed)
    __builtin_amdgcn_wave_barrier();
    if (valid && lower == 0) wave_cnt[wave][digit] = (uint16_t)(before + count_in_step);
    __builtin_amdgcn_wave_barrier();
    if (valid) {
      const uint32_t pos = local_start[digit] + before + rank_in_step;
      lds_k[pos] = (K)key;
      lds_v[pos] = val;
    }
  }
  __syncthreads();
  // ---- write out: consecutive local positions of one digit are consecutive global addresses ----
  const uint32_t count = block0 < n ? min((uint32_t)BLOCK_ITEMS, n - block0) : 0u;
  for (uint32_t pos = tid; pos < count; pos += kSortThreads) {
    const uint32_t key = (uint32_t)lds_k[pos];
    const uint32_t digit = ((key - key_base) >> shift) & mask;
    const uint32_t dst = digit_base[digit] + (pos - local_start[digit]);
    keys_out[dst] = (K)key;
    vals_out[dst] = lds_v[pos];
  }
}

// ---- inclusive scan of tiles_touched gathered through idx_sorted: block sums, then a scan in which every block first adds up the
// sums of the blocks before it (two launches) ----
// items per thread of the two scan kernels: 16 (4 096 per block), or 4 for up to 2 M items -- 245 blocks of 4 096 are ONE workgroup per CU
// for a kernel that is nothing but latency (C2: scan 0.026 -> 0.020 ms); every block adds up the sums of the blocks before it, which is
// quadratic in their number, hence the cap
constexpr int kScanItemsMax = 16;
__device__ __forceinline__ uint32_t block_exclusive_scan(uint32_t v, uint32_t* total) {
  __shared__ uint32_t ws[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  uint32_t x = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = __shfl_up(x, d);
    if (lane >= d) x += y;
  }
  __syncthreads();  // protects ws across successive calls
  if (lane == 63) ws[wave] = x;
  __syncthreads();
  uint32_t off = 0;
  for (int w = 0; w < wave; w++) off += ws[w];
  if (total) *total = ws[0] + ws[1] + ws[2] + ws[3];
  return off + x - v;
}

// PACKED: vals hold packed tile rectangles (x0 | y0<<8 | w<<16 | h<<24); the scanned quantity is w*h and `gathered` keeps the raw
// rectangles in scan order (the instance emission then reads them sequentially).
__device__ __forceinline__ uint32_t rect_count(uint32_t r) { return ((r >> 16) & 255u) * (r >> 24); }

// sq_part != nullptr: sq_part[block] = sum of (count^2) over the block's items; gather_scan_kernel's last block adds the partial sums up
// (the launcher's splat-size statistic, rg_launch.inc::use_streams -- no atomics: 1 000 same-address atomics cost 10 us here).
template <bool PACKED, int kScanItems>
__global__ void __launch_bounds__(kSortThreads) gather_block_sums_kernel(const uint32_t* __restrict__ vals, const uint32_t* __restrict__ idx,
                                                                         uint32_t n, uint32_t* __restrict__ block_sums,
                                                                         uint32_t* __restrict__ gathered, unsigned long long* __restrict__ sq_part) {
  // the block's 4096 items, striped over the threads (the sum does not care about the order): coalesced index loads and stores,
  // and all 16 dependent gathers of a thread in flight together
  const uint32_t base = blockIdx.x * (uint32_t)(kSortThreads * kScanItems) + threadIdx.x;
  uint32_t id[kScanItems], v[kScanItems];
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    const uint32_t i = base + k * kSortThreads;
    id[k] = i < n ? (idx ? idx[i] : i) : 0xFFFFFFFFu;
  }
#pragma unroll
  for (int k = 0; k < kScanItems; k++) v[k] = id[k] != 0xFFFFFFFFu ? vals[id[k]] : 0u;  // the only random gather
  uint32_t s = 0;
  unsigned long long s2 = 0ull;
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    const uint32_t i = base + k * kSortThreads;
    if (i < n) gathered[i] = v[k];   // the scan pass re-reads it sequentially
    const uint32_t c = PACKED ? rect_count(v[k]) : v[k];
    s += c;
    s2 += (unsigned long long)c * c;
  }
  if (sq_part) {
    __shared__ unsigned long long w2[4];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s2 += __shfl_xor(s2, d);
    if ((threadIdx.x & 63) == 0) w2[threadIdx.x >> 6] = s2;
    __syncthreads();
    if (threadIdx.x == 0) sq_part[blockIdx.x] = (w2[0] + w2[1]) + (w2[2] + w2[3]);
  }
  uint32_t total;
  block_exclusive_scan(s, &total);
  if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

template <bool PACKED, int kScanItems>
__global__ void __launch_bounds__(kSortThreads) gather_scan_kernel(const uint32_t* __restrict__ gathered, uint32_t n,
                                                                   const uint32_t* __restrict__ block_sums, uint32_t* __restrict__ out,
                                                                   const unsigned long long* __restrict__ sq_part,
                                                                   unsigned long long* __restrict__ sq_sum) {
  if (sq_sum && blockIdx.x == gridDim.x - 1) {   // the last block also adds up gather_block_sums_kernel's partial sums of squares
    __shared__ unsigned long long w2[4];
    unsigned long long t = 0ull;
    for (uint32_t j = threadIdx.x; j < gridDim.x; j += kSortThreads) t += sq_part[j];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) t += __shfl_xor(t, d);
    if ((threadIdx.x & 63) == 0) w2[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) *sq_sum = (w2[0] + w2[1]) + (w2[2] + w2[3]);
  }
  const uint32_t base = blockIdx.x * (uint32_t)(kSortThreads * kScanItems) + threadIdx.x * kScanItems;
  uint32_t v[kScanItems];
  uint32_t s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    const uint32_t i = base + k;
    v[k] = i < n ? (PACKED ? rect_count(gathered[i]) : gathered[i]) : 0u;
    s += v[k];
  }
  // sum of the block sums before this block: every block adds them up itself (a few hundred values) -- cheaper than a
  // one-workgroup kernel in between, which costs a launch floor of ~5 us
  uint32_t before = 0;
  for (uint32_t j = threadIdx.x; j < blockIdx.x; j += kSortThreads) before += block_sums[j];
  uint32_t before_total;
  block_exclusive_scan(before, &before_total);
  uint32_t run = before_total + block_exclusive_scan(s, nullptr);
#pragma unroll
  for (int k = 0; k < kScanItems; k++) {
    const uint32_t i = base + k;
    run += v[k];
    if (i < n) out[i] = run;  // inclusive
  }
}

}  // namespace

static int sort_items_per_thread(size_t n) {
  static const int forced = [] { const char* e = getenv("RADEGS_SORT_ITEMS"); return e ? atoi(e) : 0; }();   // 8 / 16 / 32: measurements only
  if (forced == 8 || forced == 16 || forced == 32) return forced;
  // measured (C2 1 M / 3.9 M, C4 5 M / 19.6 M, C5 50 M items): 8 up to 3 M, 16 above, 32 only for tens of millions
  return n > (size_t(32) << 20) ? 32 : (n > (size_t(3) << 20) ? 16 : 8);
}

size_t sort_temp_bytes(size_t n) {
  const size_t nblocks = (n + kSortThreads * 8 - 1) / (kSortThreads * 8);  // upper bound over both block sizes
  // ping-pong keys + values, [512][nblocks] histogram (9-bit digits; 256 rows otherwise), 512 totals
  return 2 * (n * sizeof(uint32_t) + 256) + (2 * kBins * (nblocks + 1)) * sizeof(uint32_t) + 2 * kBins * sizeof(uint32_t) + 1024;
}

// Sorts (keys_in, vals_in) by bits [0, end_bit) of the key into (keys_out, vals_out).  vals_in == nullptr means
// "values are 0..n-1".  keys_in/vals_in are left untouched; temp must hold sort_temp_bytes(n).
// n_dev != nullptr: n is a CAPACITY (it sizes the grid and the temp storage) and the number of valid items is min(n, *n_dev),
// read on the device -- the caller does not have to wait for it.
template <class K>
static hipError_t radix_sort_pairs(void* temp, size_t temp_bytes, const K* keys_in, K* keys_out, const uint32_t* vals_in,
                                   uint32_t* vals_out, size_t n, int end_bit, hipStream_t stream, const uint32_t* n_dev,
                                   uint32_t key_base = 0, int digit_bits = 8) {
  if (n == 0) return hipSuccess;
  if (temp_bytes < sort_temp_bytes(n)) return hipErrorInvalidValue;
  if (n > 0xFFFFFFFFull - 65536) return hipErrorInvalidValue;
  const int items = sort_items_per_thread(n);
  const uint32_t nblocks = (uint32_t)((n + (size_t)kSortThreads * items - 1) / ((size_t)kSortThreads * items));
  char* p = static_cast<char*>(temp);
  auto take = [&](size_t bytes) { char* r = p; p += (bytes + 255) & ~size_t(255); return r; };
  K* tkeys = reinterpret_cast<K*>(take(n * sizeof(uint32_t)));
  uint32_t* tvals = reinterpret_cast<uint32_t*>(take(n * sizeof(uint32_t)));
  const int bins = digit_bits > 8 ? 512 : 256;
  uint32_t* hist = reinterpret_cast<uint32_t*>(take((size_t)bins * nblocks * sizeof(uint32_t)));
  uint32_t* totals = reinterpret_cast<uint32_t*>(take(bins * sizeof(uint32_t)));
  const int passes = (end_bit + digit_bits - 1) / digit_bits;
  const int width = (end_bit + passes - 1) / passes;  // balanced digits: 13 bits -> 7 + 6, 32 -> 8 x 4
  const K* src_k = keys_in;
  const uint32_t* src_v = vals_in;
  for (int pass = 0; pass < passes; pass++) {
    const int shift = pass * width;
    const int nbits = (end_bit - shift) < width ? (end_bit - shift) : width;
    const uint32_t mask = (1u << nbits) - 1u;
    // destinations alternate so that the LAST pass lands in (keys_out, vals_out)
    const bool to_out = ((passes - 1 - pass) % 2) == 0;
    K* dst_k = to_out ? keys_out : tkeys;
    uint32_t* dst_v = to_out ? vals_out : tvals;
#define RG_SORT_PASS_B(I_, B_)                                                                                                             \
  do {                                                                                                                                     \
    hipLaunchKernelGGL((digit_histogram_kernel<I_, K, B_>), dim3(nblocks), dim3(kSortThreads), 0, stream, src_k, (uint32_t)n, shift, mask, \
                       hist, nblocks, n_dev, key_base);                                                                                    \
    hipLaunchKernelGGL(scan_rows_kernel, dim3(B_), dim3(kSortThreads), 0, stream, hist, nblocks, totals);                                  \
    hipLaunchKernelGGL((scatter_kernel<I_, K, B_>), dim3(nblocks), dim3(kSortThreads), 0, stream, src_k, src_v, dst_k, dst_v, (uint32_t)n, \
                       shift, mask, nbits, hist, nblocks, totals, n_dev, key_base);                                                        \
  } while (0)
#define RG_SORT_PASS(I_)                     \
  do {                                       \
    if (bins == 512) RG_SORT_PASS_B(I_, 512); \
    else RG_SORT_PASS_B(I_, 256);            \
  } while (0)
    if (items == 32) RG_SORT_PASS(32);
    else if (items == 16) RG_SORT_PASS(16);
    else RG_SORT_PASS(8);
#undef RG_SORT_PASS_B
#undef RG_SORT_PASS
    src_k = dst_k;
    src_v = dst_v;
  }
  return hipGetLastError();
}

hipError_t radix_sort_pairs_u32(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in,
                                uint32_t* vals_out, size_t n, int end_bit, hipStream_t stream, const uint32_t* n_dev) {
  return radix_sort_pairs<uint32_t>(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, end_bit, stream, n_dev);
}
// Sorts on bits [0, 27) of (key - key_base) in THREE passes of 9-bit digits (512 bins).  For keys that are the float bits of values in
// [bits^-1(key_base), ...) the caller guarantees (key - key_base) < 2^27 for every item whose order matters; other items land anywhere.
hipError_t radix_sort_pairs_u32_27(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in,
                                   uint32_t* vals_out, size_t n, uint32_t key_base, hipStream_t stream) {
  return radix_sort_pairs<uint32_t>(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, 27, stream, nullptr, key_base, 9);
}
// the same sort over 16-bit keys (end_bit <= 16)
hipError_t radix_sort_pairs_u16(void* temp, size_t temp_bytes, const uint16_t* keys_in, uint16_t* keys_out, const uint32_t* vals_in,
                                uint32_t* vals_out, size_t n, int end_bit, hipStream_t stream, const uint32_t* n_dev) {
  if (end_bit > 16) return hipErrorInvalidValue;
  return radix_sort_pairs<uint16_t>(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, n, end_bit, stream, n_dev);
}

}  // namespace rg

namespace rg {
static int scan_items_per_thread(size_t n) { return n <= (size_t(2) << 20) ? 4 : kScanItemsMax; }
size_t scan_temp_bytes(size_t n) {   // block sums (u32) + partial sums of squares (u64) per block, then the gathered copy
  const size_t per_block = (size_t)kSortThreads * scan_items_per_thread(n);
  return ((n + per_block - 1) / per_block + 64) * (sizeof(uint32_t) + sizeof(unsigned long long)) + 1024 + n * sizeof(uint32_t);
}

// out[i] = sum_{j <= i} vals[idx[j]]   (rasterizer_impl.cu:350's InclusiveSum, taken in depth order); idx == nullptr: identity
// packed_out != nullptr: `vals` are packed tile rectangles, the scan runs over their tile counts and packed_out[i] receives
// vals[idx[i]] (n words).
hipError_t inclusive_scan_gather_u32(void* temp, size_t temp_bytes, const uint32_t* vals, const uint32_t* idx, uint32_t* out, size_t n,
                                     hipStream_t stream, uint32_t* packed_out, unsigned long long* sq_sum) {
  if (n == 0) return hipSuccess;
  if (temp_bytes < scan_temp_bytes(n)) return hipErrorInvalidValue;
  const int items = scan_items_per_thread(n);
  const uint32_t nblocks = (uint32_t)((n + (size_t)kSortThreads * items - 1) / ((size_t)kSortThreads * items));
  const uint32_t nb64 = (nblocks + 64) & ~63u;
  unsigned long long* sq_part = static_cast<unsigned long long*>(temp);                // [nb64] (8-byte aligned: first in the buffer)
  uint32_t* block_sums = reinterpret_cast<uint32_t*>(sq_part + nb64);                  // [nb64]
  uint32_t* gathered = packed_out ? packed_out : block_sums + nb64;
  if (!sq_sum) sq_part = nullptr;
#define RG_SCAN_LAUNCH(P_, I_)                                                                                                                      \
  do {                                                                                                                                               \
    hipLaunchKernelGGL((gather_block_sums_kernel<P_, I_>), dim3(nblocks), dim3(kSortThreads), 0, stream, vals, idx, (uint32_t)n, block_sums, gathered, \
                       sq_part);                                                                                                                     \
    hipLaunchKernelGGL((gather_scan_kernel<P_, I_>), dim3(nblocks), dim3(kSortThreads), 0, stream, gathered, (uint32_t)n, block_sums, out, sq_part,     \
                       sq_sum);                                                                                                                      \
  } while (0)
  if (packed_out) { if (items == 4) RG_SCAN_LAUNCH(true, 4); else RG_SCAN_LAUNCH(true, kScanItemsMax); }
  else { if (items == 4) RG_SCAN_LAUNCH(false, 4); else RG_SCAN_LAUNCH(false, kScanItemsMax); }
#undef RG_SCAN_LAUNCH
  return hipGetLastError();
}
}  // namespace rg
