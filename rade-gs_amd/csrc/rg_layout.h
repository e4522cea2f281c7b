// rg_layout.h -- private layout of the three state buffers (geometry / binning / image) and of
// the per-Gaussian gradient accumulator.  The reference carves SoA arrays out of opaque byte
// buffers (DGR/cuda_rasterizer/rasterizer_impl.cu:190-250, rasterizer_impl.h:29-94); the layout
// is private to forward/backward there too (Python only sees uint8 tensors), so it is redesigned
// here for how gfx950 actually reads it:
//
//   * splat_a  [P][16] f32, one 64-byte record per Gaussian = everything the blend loops gather
//              for one list entry in depth mode, fetched as 4 x dwordx4 from ONE cache line:
//                 { mx, my, cx, cy | cz, op, thr, ts | r, g, b, rpx | rpy, nx, ny, nz }
//              (mx,my pixel centre; c* conic; op = opacity*coef; thr = conservative skip
//              threshold on the exponent; ts = |p_view|; rp = ray-space depth plane; n = normal)
//   * splat_b  [P][12] f32 (48 B), only when the coord map is requested:
//                 { cp0..cp3 | cp4, cp5, vpx, vpy | vpz, 0, 0, 0 }
//   * depth_key/tiles_touched/clamped: per-Gaussian scalars for binning and the backward
//   * the reference's cov3D array is not stored at all: the backward re-derives it from
//     scale/rotation with the same code (bit-identical), saving 48 B/Gaussian of traffic.
//
// Binning: Gaussians are radix-sorted ONCE by depth (P 32-bit keys), instances are emitted in
// that order with the tile id as key, and a stable sort on the `bit` tile bits finishes the job.
// The result is the same (tile, depth, index) order a stable 64-bit [tile|depth] sort gives
// (rasterizer_impl.cu:70-111,373-381), at ~1/4 of the sort traffic.
#pragma once
#include <stddef.h>
#include <stdint.h>

namespace rg {

constexpr size_t kAlign = 256;
inline size_t align_up(size_t v) { return (v + kAlign - 1) & ~(kAlign - 1); }

struct Carver {
  char* base;
  size_t off;
  explicit Carver(void* b) : base(static_cast<char*>(b)), off(0) {}
  template <class T> T* take(size_t count) {
    off = align_up(off);
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += count * sizeof(T);
    return p;
  }
  size_t total() const { return align_up(off) + kAlign; }
};

struct GeomState {
  float* splat_a;          // [P*16]
  float* splat_b;          // [P*12] (coord only)
  uint32_t* tiles_touched; // [P]
  uint32_t* depth_key;     // [P]
  uint8_t* clamped;        // [P]
  uint32_t* rect;          // [P] packed tile rectangle (x0 | y0<<8 | w<<16 | h<<24): all instance emission needs per Gaussian
  // forward-only scratch
  uint32_t* depth_key_sorted;  // [P]
  uint32_t* idx_sorted;        // [P]
  uint32_t* offsets;           // [P] inclusive scan of tiles_touched in depth order
  uint32_t* rect_sorted;       // [P] packed tile rectangles in depth order (written by the scan's gather, read by the emission)
  unsigned long long* tile_sq_sum;   // [1] sum over the Gaussians of (tiles touched)^2: with num_rendered the size of the splat the average
                               // instance belongs to, which chooses between the two blend formulations (rg_launch.inc::use_streams)
  char* temp;                  // device-primitive temp storage
  size_t temp_bytes;
  size_t total;
  static GeomState carve(void* buf, size_t P, bool coord, size_t temp_bytes) {
    Carver c(buf);
    GeomState g;
    g.splat_a = c.take<float>(P * 16);
    g.splat_b = c.take<float>(coord ? P * 12 : 0);
    g.tiles_touched = c.take<uint32_t>(P);
    g.depth_key = c.take<uint32_t>(P);
    g.clamped = c.take<uint8_t>(P);
    g.rect = c.take<uint32_t>(P);
    g.depth_key_sorted = c.take<uint32_t>(P);
    g.idx_sorted = c.take<uint32_t>(P);
    g.offsets = c.take<uint32_t>(P);
    g.rect_sorted = c.take<uint32_t>(P);
    g.tile_sq_sum = c.take<unsigned long long>(2);
    g.temp = c.take<char>(temp_bytes);
    g.temp_bytes = temp_bytes;
    g.total = c.total();
    return g;
  }
};

struct BinState {
  uint32_t* point_list;        // [R] sorted Gaussian indices (kept for the backward)
  uint32_t* tile_keys_sorted;  // [R]
  uint32_t* tile_keys;         // [R] unsorted
  uint32_t* point_list_unsorted;  // [R]
  char* temp;
  size_t temp_bytes;
  size_t total;
  static BinState carve(void* buf, size_t R, size_t temp_bytes) {
    Carver c(buf);
    BinState b;
    b.point_list = c.take<uint32_t>(R);
    b.tile_keys_sorted = c.take<uint32_t>(R);
    b.tile_keys = c.take<uint32_t>(R);
    b.point_list_unsorted = c.take<uint32_t>(R);
    b.temp = c.take<char>(temp_bytes);
    b.temp_bytes = temp_bytes;
    b.total = c.total();
    return b;
  }
};

// Sub-tile entry streams (rg_streams.inc): every 16x16 tile is split into 8 blocks of 8x4 pixels and every block gets its
// own exactly culled, depth-ordered sub-list of the tile's list.  Storage is an array of ROUND CHUNKS: one chunk = the 16
// entries a block's 16 lanes stage at a time = 16 x {gaussian id, position in the tile list} followed by the 16 per-lane
// contribution words the forward blend leaves for the backward (bit j / 16+j: the lane's first / second pixel blended
// entry j of this round).  The chunks of a block are consecutive and start at chunk blk_base[tile * 8 + block]: block_counts_kernel
// counts every list first and the lists are laid out back to back, sum_b ceil(n_b / 16) chunks per tile (round 5; rounds 2-4 reserved
// the tile's whole list for every block and needed no count: 8 * ((R >> 4) + tiles + 1) chunks, which stays the capacity that can never
// overflow -- stream_chunk_capacity_safe -- and is what a forward without a usage history allocates).
constexpr int kMaskShift = 24;                       // entry streams: instance value = Gaussian index | block mask << 24 while it is sorted
constexpr uint32_t kGidMask = (1u << kMaskShift) - 1u;
constexpr int kBlocksPerTile = 8;
constexpr int kListGridMax = 2048;   // workgroups of the two list-building kernels: fills the chip once (8 workgroups of 4 waves per CU)
constexpr int kChunkWords = 48;   // 16 x uint2 + 16 x u32
constexpr uint32_t kStreamTag = 0x53545247u;   // ImageState::stream_tag
inline size_t stream_chunk_capacity_safe(size_t R, size_t tiles) { return (size_t)kBlocksPerTile * ((R >> 4) + tiles + 1); }

struct ImageState {
  uint32_t* ranges;      // [2*tiles]  (start,end) per tile
  uint32_t* n_contrib;   // [2*H*W]    plane 0: last contributor, plane 1: last contributor with T>0.5
  float* accum_coord;    // [3*H*W]
  float* accum_depth;    // [H*W]
  float* normal_length;  // [H*W]
  // entry streams (only when stream_chunks != 0).  They live at the END of the image state so that every offset above and the
  // bases below depend on (W, H) alone: the backward finds them without knowing the capacity the forward allocated for.
  uint32_t* blk_count;     // [8*tiles] entries in each block's list
  uint32_t* blk_base;      // [8*tiles] first chunk of each block's list
  uint32_t* wg_need;       // [kListGridMax] chunks the tiles of each list-building workgroup need (block_counts_kernel -> block_lists_kernel)
  uint32_t* blk_consumed;  // [8*tiles] entries of it the forward walked before all of the block's pixels had terminated
  uint32_t* blk_order;     // [8*tiles] block ids (tile*8 + block) in wave order: wave w of the blend kernels walks entries 4w..4w+3
  uint32_t* stream_tag;    // [4] word 0: kStreamTag when THIS forward wrote entry streams into this buffer, 0 when it did not (written by
                           // the instance emission of every forward; the stream backward kernels refuse any other value); word 1: chunks
                           // the forward's lists need in all (block_lists_kernel; also when they exceed the capacity, so the host learns the
                           // real need); word 3: set when they did not fit
  uint32_t* blk_chunks;    // [stream_chunks * kChunkWords]
  size_t total;
  static ImageState carve(void* buf, size_t W, size_t H, size_t stream_chunks = 0) {
    Carver c(buf);
    ImageState s;
    const size_t tiles = ((W + 15) / 16) * ((H + 15) / 16), N = W * H;
    s.ranges = c.take<uint32_t>(2 * tiles);
    s.n_contrib = c.take<uint32_t>(2 * N);
    s.accum_coord = c.take<float>(3 * N);
    s.accum_depth = c.take<float>(N);
    s.normal_length = c.take<float>(N);
    s.blk_count = c.take<uint32_t>(kBlocksPerTile * tiles);
    s.blk_base = c.take<uint32_t>(kBlocksPerTile * tiles);
    s.wg_need = c.take<uint32_t>(kListGridMax);
    s.blk_consumed = c.take<uint32_t>(kBlocksPerTile * tiles);
    s.blk_order = c.take<uint32_t>(kBlocksPerTile * tiles);
    s.stream_tag = c.take<uint32_t>(4);
    s.blk_chunks = c.take<uint32_t>(stream_chunks * kChunkWords);
    s.total = c.total();
    return s;
  }
};

// State of the integrate() path beyond the three buffers above (the reference's fourth and fifth resizable buffers,
// rasterize_points.cu:303-309: PointState / point binning): the INTE preprocess record, the projected query points and
// their per-PIXEL bins.
struct PointState {
  float* inte_rec;       // [P*8]  {icr0..icr5, well, 0}: inverse ray-space covariance (upper triangle) + conditioning flag
  float* p2d;            // [PN*2] projected position (pixels)
  float* pdepth;         // [PN]   |p_view|
  uint32_t* ppix;        // [PN]   pixel index or 0xFFFFFFFF (not projected)
  uint32_t* pt_sorted;   // [PN]   point ids grouped by pixel
  uint32_t* pix_count;   // [H*W]
  uint32_t* pix_incl;    // [H*W]  inclusive scan of pix_count
  float* final_T;        // [H*W]
  char* temp;
  size_t temp_bytes;
  size_t total;
  static PointState carve(void* buf, size_t P, size_t PN, size_t HW, size_t temp_bytes) {
    Carver c(buf);
    PointState s;
    s.inte_rec = c.take<float>(P * 8);
    s.p2d = c.take<float>(PN * 2);
    s.pdepth = c.take<float>(PN);
    s.ppix = c.take<uint32_t>(PN);
    s.pt_sorted = c.take<uint32_t>(PN);
    s.pix_count = c.take<uint32_t>(HW);
    s.pix_incl = c.take<uint32_t>(HW);
    s.final_T = c.take<float>(HW);
    s.temp = c.take<char>(temp_bytes);
    s.temp_bytes = temp_bytes;
    s.total = c.total();
    return s;
  }
};

// Per-Gaussian gradient accumulator record written by the blend backward with one 16/25-lane
// atomic instruction: floats in rg::SplatAcc order; 16 per Gaussian without the coord map
// (64 B = one line), 32 with it.
inline int acc_record_floats(bool coord) { return coord ? 32 : 16; }

}  // namespace rg
