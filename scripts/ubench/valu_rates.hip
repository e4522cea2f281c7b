// VALU issue-rate micro-benchmark for gfx950 (measurement tool, not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates scripts/ubench/valu_rates.hip && /tmp/valu_rates
// Every kernel runs a loop of 64 copies of ONE instruction pattern on independent registers; the grid fills every SIMD with
// `waves` resident waves.  Reported: SIMD cycles per wave64 instruction at the measured clock (wall time x nominal clock / count).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string>
#include <vector>

#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

#define KERNEL(name, body_asm, clobber_setup)                                                              \
  __global__ void __launch_bounds__(64) name(float* out, int iters) {                                      \
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    float b0 = 1.0001f, b1 = 0.9999f;                                                                      \
    for (int i = 0; i < iters; i++) {                                                                      \
      asm volatile(REP8(body_asm)                                                                          \
                   : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)        \
                   : "v"(b0), "v"(b1)                                                                      \
                   : "vcc", "s20", "s21");                                                                   \
    }                                                                                                      \
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 123.456f) out[0] = a0;                                    \
  }

// each body: 8 instructions on a0..a7 (independent), so REP8 -> 64 instructions per loop iteration
KERNEL(k_fma3, "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n", 0)
KERNEL(k_fmac, "v_fmac_f32 %0, %8, %9\n v_fmac_f32 %1, %8, %9\n v_fmac_f32 %2, %8, %9\n v_fmac_f32 %3, %8, %9\n v_fmac_f32 %4, %8, %9\n v_fmac_f32 %5, %8, %9\n v_fmac_f32 %6, %8, %9\n v_fmac_f32 %7, %8, %9\n", 0)
KERNEL(k_mul, "v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n", 0)
KERNEL(k_add, "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n", 0)
typedef float f2v __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(64) k_pkfma(float* out, int iters) {
  f2v a0 = {(float)threadIdx.x, 1.f}, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
  f2v b0 = {1.0001f, 0.9999f}, b1 = {0.9999f, 1.0001f};
  for (int i = 0; i < iters; i++) {
    asm volatile(REP8("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n")
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : "v"(b0), "v"(b1));
  }
  f2v t = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
  if (t.x + t.y == 123.456f) out[0] = t.x;
}
KERNEL(k_cnd32, "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n", 0)
KERNEL(k_cnd64, "v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]\n", 0)
KERNEL(k_adddpp, "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n", 0)
KERNEL(k_addror, "v_add_f32_dpp %0, %0, %0 row_ror:8 row_mask:0xf bank_mask:0x3\n v_add_f32_dpp %1, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n v_add_f32_dpp %2, %2, %2 row_ror:8 row_mask:0xf bank_mask:0x3\n v_add_f32_dpp %3, %3, %3 row_ror:8 row_mask:0xf bank_mask:0x3\n v_add_f32_dpp %4, %4, %4 row_ror:8 row_mask:0xf bank_mask:0x3\n v_add_f32_dpp %5, %5, %5 row_ror:8 row_mask:0xf bank_mask:0x3\n v_add_f32_dpp %6, %6, %6 row_ror:8 row_mask:0xf bank_mask:0x3\n v_add_f32_dpp %7, %7, %7 row_ror:8 row_mask:0xf bank_mask:0x3\n", 0)
KERNEL(k_exp, "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n", 0)
KERNEL(k_rcp, "v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n", 0)
KERNEL(k_cmp64, "v_cmp_lt_f32_e64 s[20:21], %0, %8\n v_cmp_lt_f32_e64 s[20:21], %1, %8\n v_cmp_lt_f32_e64 s[20:21], %2, %8\n v_cmp_lt_f32_e64 s[20:21], %3, %8\n v_cmp_lt_f32_e64 s[20:21], %4, %8\n v_cmp_lt_f32_e64 s[20:21], %5, %8\n v_cmp_lt_f32_e64 s[20:21], %6, %8\n v_cmp_lt_f32_e64 s[20:21], %7, %8\n", 0)
KERNEL(k_min, "v_min_f32 %0, %0, %8\n v_min_f32 %1, %1, %8\n v_min_f32 %2, %2, %8\n v_min_f32 %3, %3, %8\n v_min_f32 %4, %4, %8\n v_min_f32 %5, %5, %8\n v_min_f32 %6, %6, %8\n v_min_f32 %7, %7, %8\n", 0)
KERNEL(k_lshlor, "v_lshl_or_b32 %0, %0, 1, %8\n v_lshl_or_b32 %1, %1, 1, %8\n v_lshl_or_b32 %2, %2, 1, %8\n v_lshl_or_b32 %3, %3, 1, %8\n v_lshl_or_b32 %4, %4, 1, %8\n v_lshl_or_b32 %5, %5, 1, %8\n v_lshl_or_b32 %6, %6, 1, %8\n v_lshl_or_b32 %7, %7, 1, %8\n", 0)
// a dependent chain: each instruction consumes the previous result (one register)
KERNEL(k_fma_dep, "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %0, %0, %8, %9\n", 0)
KERNEL(k_fma_dep2, "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n", 0)

KERNEL(k_cnd32_b, "v_cndmask_b32 %0, %8, %9, vcc\n v_cndmask_b32 %1, %8, %9, vcc\n v_cndmask_b32 %2, %8, %9, vcc\n v_cndmask_b32 %3, %8, %9, vcc\n v_cndmask_b32 %4, %8, %9, vcc\n v_cndmask_b32 %5, %8, %9, vcc\n v_cndmask_b32 %6, %8, %9, vcc\n v_cndmask_b32 %7, %8, %9, vcc\n ", 0)
KERNEL(k_cmp_cnd32, "v_cmp_lt_f32 vcc, %8, %9\n v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n", 0)
KERNEL(k_cmp32, "v_cmp_lt_f32 vcc, %0, %8\n v_cmp_lt_f32 vcc, %1, %8\n v_cmp_lt_f32 vcc, %2, %8\n v_cmp_lt_f32 vcc, %3, %8\n v_cmp_lt_f32 vcc, %4, %8\n v_cmp_lt_f32 vcc, %5, %8\n v_cmp_lt_f32 vcc, %6, %8\n v_cmp_lt_f32 vcc, %7, %8\n ", 0)
KERNEL(k_movdpp, "v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n ", 0)
KERNEL(k_bfe, "v_bfe_i32 %0, %0, 3, 1\n v_bfe_i32 %1, %1, 3, 1\n v_bfe_i32 %2, %2, 3, 1\n v_bfe_i32 %3, %3, 3, 1\n v_bfe_i32 %4, %4, 3, 1\n v_bfe_i32 %5, %5, 3, 1\n v_bfe_i32 %6, %6, 3, 1\n v_bfe_i32 %7, %7, 3, 1\n ", 0)
KERNEL(k_and, "v_and_b32 %0, %0, %8\n v_and_b32 %1, %1, %8\n v_and_b32 %2, %2, %8\n v_and_b32 %3, %3, %8\n v_and_b32 %4, %4, %8\n v_and_b32 %5, %5, %8\n v_and_b32 %6, %6, %8\n v_and_b32 %7, %7, %8\n ", 0)
KERNEL(k_mov, "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n ", 0)
KERNEL(k_rndne, "v_rndne_f32 %0, %0\n v_rndne_f32 %1, %1\n v_rndne_f32 %2, %2\n v_rndne_f32 %3, %3\n v_rndne_f32 %4, %4\n v_rndne_f32 %5, %5\n v_rndne_f32 %6, %6\n v_rndne_f32 %7, %7\n ", 0)
KERNEL(k_cvti, "v_cvt_i32_f32 %0, %0\n v_cvt_i32_f32 %1, %1\n v_cvt_i32_f32 %2, %2\n v_cvt_i32_f32 %3, %3\n v_cvt_i32_f32 %4, %4\n v_cvt_i32_f32 %5, %5\n v_cvt_i32_f32 %6, %6\n v_cvt_i32_f32 %7, %7\n ", 0)
KERNEL(k_lshladd, "v_lshl_add_u32 %0, %0, 3, %8\n v_lshl_add_u32 %1, %1, 3, %8\n v_lshl_add_u32 %2, %2, 3, %8\n v_lshl_add_u32 %3, %3, 3, %8\n v_lshl_add_u32 %4, %4, 3, %8\n v_lshl_add_u32 %5, %5, 3, %8\n v_lshl_add_u32 %6, %6, 3, %8\n v_lshl_add_u32 %7, %7, 3, %8\n ", 0)
KERNEL(k_fmaak, "v_fmaak_f32 %0, %0, %8, 0x3c088908\n v_fmaak_f32 %1, %1, %8, 0x3c088908\n v_fmaak_f32 %2, %2, %8, 0x3c088908\n v_fmaak_f32 %3, %3, %8, 0x3c088908\n v_fmaak_f32 %4, %4, %8, 0x3c088908\n v_fmaak_f32 %5, %5, %8, 0x3c088908\n v_fmaak_f32 %6, %6, %8, 0x3c088908\n v_fmaak_f32 %7, %7, %8, 0x3c088908\n ", 0)
KERNEL(k_sub, "v_sub_f32 %0, 1.0, %0\n v_sub_f32 %1, 1.0, %1\n v_sub_f32 %2, 1.0, %2\n v_sub_f32 %3, 1.0, %3\n v_sub_f32 %4, 1.0, %4\n v_sub_f32 %5, 1.0, %5\n v_sub_f32 %6, 1.0, %6\n v_sub_f32 %7, 1.0, %7\n ", 0)
KERNEL(k_max, "v_max_f32 %0, %0, %8\n v_max_f32 %1, %1, %8\n v_max_f32 %2, %2, %8\n v_max_f32 %3, %3, %8\n v_max_f32 %4, %4, %8\n v_max_f32 %5, %5, %8\n v_max_f32 %6, %6, %8\n v_max_f32 %7, %7, %8\n ", 0)
KERNEL(k_mul64, "v_mul_f32_e64 %0, %0, -%8\n v_mul_f32_e64 %1, %1, -%8\n v_mul_f32_e64 %2, %2, -%8\n v_mul_f32_e64 %3, %3, -%8\n v_mul_f32_e64 %4, %4, -%8\n v_mul_f32_e64 %5, %5, -%8\n v_mul_f32_e64 %6, %6, -%8\n v_mul_f32_e64 %7, %7, -%8\n ", 0)
KERNEL(k_perm32, "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n", 0)
KERNEL(k_perm16, "v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n v_permlane16_swap_b32 %0, %1\n v_permlane16_swap_b32 %2, %3\n v_permlane16_swap_b32 %4, %5\n v_permlane16_swap_b32 %6, %7\n", 0)
KERNEL(k_swz, "ds_swizzle_b32 %0, %0 offset:swizzle(SWAP,1)\n ds_swizzle_b32 %1, %1 offset:swizzle(SWAP,1)\n ds_swizzle_b32 %2, %2 offset:swizzle(SWAP,1)\n ds_swizzle_b32 %3, %3 offset:swizzle(SWAP,1)\n ds_swizzle_b32 %4, %4 offset:swizzle(SWAP,1)\n ds_swizzle_b32 %5, %5 offset:swizzle(SWAP,1)\n ds_swizzle_b32 %6, %6 offset:swizzle(SWAP,1)\n ds_swizzle_b32 %7, %7 offset:swizzle(SWAP,1)\n ", 0)
KERNEL(k_cnd64vcc, "v_cndmask_b32_e64 %0, %0, %8, vcc\n v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32_e64 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, vcc\n v_cndmask_b32_e64 %4, %4, %8, vcc\n v_cndmask_b32_e64 %5, %5, %8, vcc\n v_cndmask_b32_e64 %6, %6, %8, vcc\n v_cndmask_b32_e64 %7, %7, %8, vcc\n ", 0)
KERNEL(k_cnd32_init, "s_mov_b64 vcc, 0x5555\n v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc\n", 0)
KERNEL(k_addc, "v_addc_co_u32 %0, vcc, %0, %8, vcc\n v_addc_co_u32 %1, vcc, %1, %8, vcc\n v_addc_co_u32 %2, vcc, %2, %8, vcc\n v_addc_co_u32 %3, vcc, %3, %8, vcc\n v_addc_co_u32 %4, vcc, %4, %8, vcc\n v_addc_co_u32 %5, vcc, %5, %8, vcc\n v_addc_co_u32 %6, vcc, %6, %8, vcc\n v_addc_co_u32 %7, vcc, %7, %8, vcc\n ", 0)
KERNEL(k_cnd_mix, "v_cndmask_b32 %0, %0, %8, vcc\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n", 0)
KERNEL(k_cnd_g1, "v_cndmask_b32 %0, %0, %8, vcc\n v_fma_f32 %1, %1, %8, %9\n v_cndmask_b32 %2, %2, %8, vcc\n v_fma_f32 %3, %3, %8, %9\n v_cndmask_b32 %4, %4, %8, vcc\n v_fma_f32 %5, %5, %8, %9\n v_cndmask_b32 %6, %6, %8, vcc\n v_fma_f32 %7, %7, %8, %9\n", 0)
KERNEL(k_cnd_g3, "v_cndmask_b32 %0, %0, %8, vcc\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_cndmask_b32 %4, %4, %8, vcc\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n", 0)
KERNEL(k_cnd_2, "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n", 0)
KERNEL(k_cnd_mixenc, "v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32_e64 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32_e64 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32_e64 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32_e64 %7, %7, %8, vcc\n", 0)
typedef void (*kern_t)(float*, int);
struct Test { const char* name; kern_t k; };

int main() {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const double clock_ghz = p.clockRate * 1e-6;
  const int cus = p.multiProcessorCount;
  printf("%s: %d CUs, nominal %.2f GHz\n", p.name, cus, clock_ghz);
  float* out;
  hipMalloc(&out, 4);
  std::vector<Test> tests = {{"v_fma_f32 (3 vgpr)", k_fma3}, {"v_fmac_f32", k_fmac}, {"v_mul_f32", k_mul}, {"v_add_f32", k_add},
                             {"v_pk_fma_f32", k_pkfma}, {"v_cndmask e32 vcc", k_cnd32}, {"v_cndmask e64 sgpr", k_cnd64},
                             {"v_add_f32_dpp quad_perm", k_adddpp}, {"v_add_f32_dpp row_ror bank_mask", k_addror},
                             {"v_exp_f32", k_exp}, {"v_rcp_f32", k_rcp}, {"v_cmp_lt_f32 e64", k_cmp64}, {"v_min_f32", k_min},
                             {"v_lshl_or_b32", k_lshlor}, {"v_fma_f32 dependent chain", k_fma_dep}, {"v_fma_f32 2 chains", k_fma_dep2},
  {"v_cndmask e32 vcc dst!=src", k_cnd32_b}, {"v_cmp vcc + 7 cndmask e32", k_cmp_cnd32}, {"v_cmp_lt_f32 e32 vcc", k_cmp32}, {"v_mov_b32_dpp", k_movdpp},
  {"v_bfe_i32", k_bfe}, {"v_and_b32", k_and}, {"v_mov_b32", k_mov}, {"v_rndne_f32", k_rndne}, {"v_cvt_i32_f32", k_cvti}, {"v_lshl_add_u32", k_lshladd},
  {"v_fmaak_f32", k_fmaak}, {"v_sub_f32", k_sub}, {"v_max_f32", k_max}, {"v_mul_f32_e64 neg", k_mul64}, {"v_permlane32_swap", k_perm32}, {"v_permlane16_swap", k_perm16},
  {"ds_swizzle_b32 (+waitcnt at end)", k_swz},
 {"v_cndmask e64 with vcc operand", k_cnd64vcc}, {"s_mov vcc + 8 cndmask e32 (9 instr)", k_cnd32_init}, {"v_addc_co_u32 vcc", k_addc}, {"1 cndmask e32 + 7 fma", k_cnd_mix}, {"(cnd e32, fma) x4", k_cnd_g1}, {"(cnd e32, 3 fma) x2", k_cnd_g3}, {"2 cnd e32 + 6 fma", k_cnd_2}, {"(cnd e32, cnd e64 vcc) x4", k_cnd_mixenc}};
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int waves : {4}) {
    printf("-- %d wave(s) per SIMD\n", waves);
    for (auto& t : tests) {
      const int blocks = cus * 4 * waves;   // 64-thread blocks: one wave each
      hipLaunchKernelGGL(t.k, dim3(blocks), dim3(64), 0, 0, out, 10);
      hipDeviceSynchronize();
      hipEventRecord(e0);
      hipLaunchKernelGGL(t.k, dim3(blocks), dim3(64), 0, 0, out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double inst_per_simd = (double)iters * 64.0 * waves;
      printf("   %-34s %7.3f ms  -> %.2f SIMD-cycles per wave64 instruction (at nominal clock)\n", t.name, ms,
             ms * 1e-3 * clock_ghz * 1e9 / inst_per_simd);
    }
  }
  return 0;
}
