// radegs_prims.hip -- scan / radix sort for the binning stage, on rocPRIM.
//
// TEST-ONLY CROSS-CHECK of the hand-written primitives (radegs_sort.hip).  It is built into its own shared object
// (libradegs_prims_check.so, rade-gs_amd/build.py) and never linked into libradegs_hip.so: the product library loads it with
// dlopen only when RADEGS_PRIMS=rocprim asks for the cross-check (rg_launch.inc: PrimsCheck), so the 7 MB of rocPRIM
// instantiations stay out of what ships.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <stdint.h>

namespace {
struct GatherTiles {
  const uint32_t* tiles;
  __host__ __device__ uint32_t operator()(uint32_t idx) const { return tiles[idx]; }
};
}  // namespace

extern "C" {

size_t radegs_prims_temp_bytes_geom(size_t P) {
  size_t a = 0, b = 0;
  uint32_t* nk = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, a, nk, nk, rocprim::counting_iterator<uint32_t>(0), nk, P, 0, 32, hipStream_t(0));
  auto it = rocprim::make_transform_iterator(nk, GatherTiles{nullptr});
  (void)rocprim::inclusive_scan(nullptr, b, it, nk, P, rocprim::plus<uint32_t>(), hipStream_t(0));
  return (a > b ? a : b) + 256;
}

size_t radegs_prims_temp_bytes_bin(size_t R, int tile_bits) {
  size_t a = 0;
  uint32_t* nk = nullptr;
  (void)rocprim::radix_sort_pairs(nullptr, a, nk, nk, nk, nk, R, 0, (unsigned)tile_bits, hipStream_t(0));
  return a + 256;
}

hipError_t radegs_prims_sort_by_depth(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, uint32_t* idx_out, size_t P,
                         hipStream_t stream) {
  return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, rocprim::counting_iterator<uint32_t>(0), idx_out, P, 0, 32,
                                   stream);
}

hipError_t radegs_prims_scan_tiles_in_depth_order(void* temp, size_t temp_bytes, const uint32_t* tiles_touched, const uint32_t* idx_sorted,
                                     uint32_t* offsets, size_t P, hipStream_t stream) {
  auto it = rocprim::make_transform_iterator(idx_sorted, GatherTiles{tiles_touched});
  return rocprim::inclusive_scan(temp, temp_bytes, it, offsets, P, rocprim::plus<uint32_t>(), stream);
}

hipError_t radegs_prims_sort_by_tile(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in,
                        uint32_t* vals_out, size_t R, int tile_bits, hipStream_t stream) {
  return rocprim::radix_sort_pairs(temp, temp_bytes, keys_in, keys_out, vals_in, vals_out, R, 0, (unsigned)tile_bits, stream);
}

}  // extern "C"
