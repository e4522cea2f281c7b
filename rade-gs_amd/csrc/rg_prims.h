// rg_prims.h -- device-wide scan / radix-sort entry points used by the binning stage: the hand-written ones of radegs_sort.hip
// (the counterpart of the CUB calls at DGR/cuda_rasterizer/rasterizer_impl.cu:350,376-381) and the signatures of their
// rocPRIM cross-check.
#pragma once
#include <hip/hip_runtime_api.h>
#include <stddef.h>
#include <stdint.h>

namespace rg {

// ---- rocPRIM cross-check (radegs_prims.hip -> libradegs_prims_check.so, test-only, loaded on demand with dlopen) ----
// temp-storage requirement of the two P-sized primitives (depth sort, scan) / the R-sized one
typedef size_t (*prims_temp_geom_fn)(size_t P);
typedef size_t (*prims_temp_bin_fn)(size_t R, int tile_bits);
// (depth_key[i], i) sorted by key, stable -> keys_out, idx_out
typedef hipError_t (*prims_sort_by_depth_fn)(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, uint32_t* idx_out,
                                             size_t P, hipStream_t stream);
// offsets[i] = inclusive sum of tiles_touched[idx_sorted[j]], j <= i
typedef hipError_t (*prims_scan_fn)(void* temp, size_t temp_bytes, const uint32_t* tiles_touched, const uint32_t* idx_sorted,
                                    uint32_t* offsets, size_t P, hipStream_t stream);
// stable sort of (tile, gaussian) pairs on the low `tile_bits` bits of the tile id
typedef hipError_t (*prims_sort_by_tile_fn)(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out,
                                            const uint32_t* vals_in, uint32_t* vals_out, size_t R, int tile_bits, hipStream_t stream);

// ---- hand-written primitives (radegs_sort.hip) ----
size_t sort_temp_bytes(size_t n);
size_t scan_temp_bytes(size_t n);
hipError_t radix_sort_pairs_u32(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in,
                                uint32_t* vals_out, size_t n, int end_bit, hipStream_t stream, const uint32_t* n_dev = nullptr);
hipError_t radix_sort_pairs_u32_27(void* temp, size_t temp_bytes, const uint32_t* keys_in, uint32_t* keys_out, const uint32_t* vals_in,
                                   uint32_t* vals_out, size_t n, uint32_t key_base, hipStream_t stream);
hipError_t radix_sort_pairs_u16(void* temp, size_t temp_bytes, const uint16_t* keys_in, uint16_t* keys_out, const uint32_t* vals_in,
                                uint32_t* vals_out, size_t n, int end_bit, hipStream_t stream, const uint32_t* n_dev = nullptr);
hipError_t inclusive_scan_gather_u32(void* temp, size_t temp_bytes, const uint32_t* vals, const uint32_t* idx, uint32_t* out, size_t n,
                                     hipStream_t stream, uint32_t* packed_out = nullptr, unsigned long long* sq_sum = nullptr);

}  // namespace rg
