// Micro-benchmark: issue rate of the integer VALU ops the LDPC check-node kernel is made of (gfx950).
// Each kernel runs ITER iterations of 8 independent chains of one op; reports cycles per wave-instruction
// per SIMD assuming waves are spread evenly (4 SIMDs/CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 4096
#define DEF(name, ASM)                                                                          \
    __global__ __launch_bounds__(256) void k_##name(uint32_t* out, uint32_t seed) {             \
        uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
        uint32_t b = seed * 77 + 5, c = seed + 9;                                               \
        for (int i = 0; i < ITER; i++) {                                                        \
            asm volatile(ASM "\n" : "+v"(a0) : "v"(b), "v"(c)); asm volatile(ASM "\n" : "+v"(a1) : "v"(b), "v"(c)); \
            asm volatile(ASM "\n" : "+v"(a2) : "v"(b), "v"(c)); asm volatile(ASM "\n" : "+v"(a3) : "v"(b), "v"(c)); \
            asm volatile(ASM "\n" : "+v"(a4) : "v"(b), "v"(c)); asm volatile(ASM "\n" : "+v"(a5) : "v"(b), "v"(c)); \
            asm volatile(ASM "\n" : "+v"(a6) : "v"(b), "v"(c)); asm volatile(ASM "\n" : "+v"(a7) : "v"(b), "v"(c)); \
        }                                                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;     \
    }
DEF(add_u32, "v_add_u32 %0, %0, %1")
DEF(sub_u32, "v_sub_u32 %0, %0, %1")
DEF(xor_b32, "v_xor_b32 %0, %0, %1")
DEF(min_u32, "v_min_u32 %0, %0, %1")
DEF(med3_i32, "v_med3_i32 %0, %0, %1, %2")
DEF(min3_u32, "v_min3_u32 %0, %0, %1, %2")
DEF(sad_u16, "v_sad_u16 %0, %0, %1, %2")
DEF(sad_u8, "v_sad_u8 %0, %0, %1, %2")
DEF(bfe_u32, "v_bfe_u32 %0, %0, %1, %2")
DEF(perm_b32, "v_perm_b32 %0, %0, %1, %2")
DEF(mad_i24, "v_mad_i32_i24 %0, %0, %1, %2")
DEF(mul_i24, "v_mul_i32_i24 %0, %0, %1")
DEF(lshl_or, "v_lshl_or_b32 %0, %0, %1, %2")
DEF(add3, "v_add3_u32 %0, %0, %1, %2")
DEF(xad, "v_xad_u32 %0, %0, %1, %2")
DEF(ashr, "v_ashrrev_i32 %0, 31, %0")
DEF(pk_add_i16, "v_pk_add_i16 %0, %0, %1")
DEF(pk_add_i16_clamp, "v_pk_add_i16 %0, %0, %1 clamp")
DEF(pk_sub_i16, "v_pk_sub_i16 %0, %0, %1")
DEF(pk_min_i16, "v_pk_min_i16 %0, %0, %1")
DEF(pk_max_i16, "v_pk_max_i16 %0, %0, %1")
DEF(pk_min_u16, "v_pk_min_u16 %0, %0, %1")
DEF(pk_mad_i16, "v_pk_mad_i16 %0, %0, %1, %2")
DEF(pk_mul_lo_u16, "v_pk_mul_lo_u16 %0, %0, %1")
DEF(pk_ashr_i16, "v_pk_ashrrev_i16 %0, 15, %0")
DEF(pk_lshl_b16, "v_pk_lshlrev_b16 %0, 1, %0")
DEF(cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEF(cmp_eq, "v_cmp_eq_u32 vcc, %0, %1")
DEF(and_or, "v_and_or_b32 %0, %0, %1, %2")
DEF(bfi, "v_bfi_b32 %0, %0, %1, %2")
DEF(fma_f32, "v_fma_f32 %0, %0, %1, %2")
DEF(pk_fma_f16, "v_pk_fma_f16 %0, %0, %1, %2")
DEF(pk_max_f16, "v_pk_max_f16 %0, %0, %1")
DEF(mov_dpp, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf")

DEF(min_f32, "v_min_f32 %0, %0, %1")
DEF(max_f32, "v_max_f32 %0, %0, %1")
DEF(med3_f32, "v_med3_f32 %0, %0, %1, %2")
DEF(min3_f32, "v_min3_f32 %0, %0, %1, %2")
DEF(add_f32, "v_add_f32 %0, %0, %1")
DEF(add_f32_abs, "v_add_f32 %0, |%0|, %1")
DEF(sub_f32, "v_sub_f32 %0, %0, %1")
DEF(mul_f32, "v_mul_f32 %0, %0, %1")
DEF(cvt_f32_ubyte1, "v_cvt_f32_ubyte1 %0, %0")
DEF(cvt_pk_u8_f32, "v_cvt_pk_u8_f32 %0, %0, 1, %1")
DEF(cvt_u32_f32, "v_cvt_u32_f32 %0, %0")
DEF(cvt_f32_i32, "v_cvt_f32_i32 %0, %0")
DEF(cmp_eq_f32_cnd, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
DEF(cmp_eq_u32_cnd, "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
DEF(cmp_sgpr_cnd, "v_cmp_lt_u32 s[20:21], %0, %1\n v_cndmask_b32 %0, %0, %2, s[20:21]")
DEF(and_b32, "v_and_b32 %0, %0, %1")
DEF(or_b32, "v_or_b32 %0, %0, %1")
DEF(not_b32, "v_not_b32 %0, %0")
DEF(min_i32, "v_min_i32 %0, %0, %1")
DEF(max_i32, "v_max_i32 %0, %0, %1")
DEF(lshlrev, "v_lshlrev_b32 %0, 3, %0")
DEF(lshrrev, "v_lshrrev_b32 %0, 3, %0")
DEF(sub_sdwa, "v_sub_u32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
DEF(and_sdwa, "v_and_b32_sdwa %0, %0, %1 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_2")
DEF(mov_b32, "v_mov_b32 %0, %1")
DEF(alignbit, "v_alignbit_b32 %0, %0, %1, %2")
DEF(fma_mix, "v_fma_f32 %0, %0, %1, 1.0")
DEF(subrev_f32_neg, "v_sub_f32 %0, -%0, %1")
DEF(sub_i32_clamp, "v_sub_i32 %0, %0, %1 clamp")
DEF(add_i32_clamp, "v_add_i32 %0, %0, %1 clamp")
DEF(sub_u32_clamp, "v_sub_u32_e64 %0, %0, %1 clamp")
DEF(add_u32_e64, "v_add_u32_e64 %0, %0, %1")
DEF(lshl_sdwa, "v_lshlrev_b32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1")
DEF(med3_u32, "v_med3_u32 %0, %0, %1, %2")
DEF(max3_u32, "v_max3_u32 %0, %0, %1, %2")
DEF(add_i16_clamp, "v_add_i16 %0, %0, %1 clamp")
DEF(sub_co, "v_sub_co_u32 %0, vcc, %0, %1")
DEF(subb_co, "v_subbrev_co_u32 %0, vcc, 0, %0, vcc")
DEF(xor3, "v_bitop3_b32 %0, %0, %1, %2 bitop3:0x96")
DEF(lshl_add, "v_lshl_add_u32 %0, %0, 3, %1")
DEF(add_lshl, "v_add_lshl_u32 %0, %0, %1, 3")
DEF(or3, "v_or3_b32 %0, %0, %1, %2")
// 64-bit register pairs: packed fp32 (two fp32 operations per lane and instruction), 64-bit shifts / moves
#define DEF2(name, ASM)                                                                         \
    __global__ __launch_bounds__(256) void k_##name(uint32_t* out, uint32_t seed) {             \
        double a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19; \
        double b = seed * 77 + 5, c = seed + 9;                                                 \
        for (int i = 0; i < ITER; i++) {                                                        \
            asm volatile(ASM "\n" : "+v"(a0) : "v"(b), "v"(c)); asm volatile(ASM "\n" : "+v"(a1) : "v"(b), "v"(c)); \
            asm volatile(ASM "\n" : "+v"(a2) : "v"(b), "v"(c)); asm volatile(ASM "\n" : "+v"(a3) : "v"(b), "v"(c)); \
            asm volatile(ASM "\n" : "+v"(a4) : "v"(b), "v"(c)); asm volatile(ASM "\n" : "+v"(a5) : "v"(b), "v"(c)); \
            asm volatile(ASM "\n" : "+v"(a6) : "v"(b), "v"(c)); asm volatile(ASM "\n" : "+v"(a7) : "v"(b), "v"(c)); \
        }                                                                                       \
        out[blockIdx.x * blockDim.x + threadIdx.x] = (uint32_t)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7); \
    }
DEF2(pk_add_f32, "v_pk_add_f32 %0, %0, %1")
DEF2(pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
DEF2(pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2")
DEF2(pk_mov_b32, "v_pk_mov_b32 %0, %0, %1")
DEF2(lshlrev_b64, "v_lshlrev_b64 %0, 3, %0")
DEF(cvt_f32_ubyte0, "v_cvt_f32_ubyte0 %0, %0")
DEF(cvt_i32_f32, "v_cvt_i32_f32 %0, %0")
DEF(med3_f32_abs, "v_med3_f32 %0, |%0|, %1, %2")
DEF(min3_f32_abs, "v_min3_f32 %0, |%0|, |%1|, |%2|")
DEF(max3_f32, "v_max3_f32 %0, %0, %1, %2")
DEF(add_f32_clamp, "v_add_f32_e64 %0, %0, %1 clamp")
DEF(fma_f32_clamp, "v_fma_f32 %0, %0, %1, %2 clamp")
DEF(mul_legacy, "v_mul_legacy_f32 %0, %0, %1")
DEF(ldexp_f32, "v_ldexp_f32 %0, %0, %1")
DEF(cvt_pk_i16_i32, "v_cvt_pk_i16_i32 %0, %0, %1")
DEF(cvt_pk_u16_u32, "v_cvt_pk_u16_u32 %0, %0, %1")
DEF(cvt_pknorm_i16, "v_cvt_pknorm_i16_f32 %0, %0, %1")
DEF(sat_pk_u8_i16, "v_sat_pk_u8_i16 %0, %0")
DEF(pk_sub_i16_clamp, "v_pk_sub_i16 %0, %0, %1 clamp")
DEF(pk_sub_u16_clamp, "v_pk_sub_u16 %0, %0, %1 clamp")
DEF(pk_add_u16_clamp, "v_pk_add_u16 %0, %0, %1 clamp")
DEF(pk_max_u16, "v_pk_max_u16 %0, %0, %1")
DEF(min_u16, "v_min_u16 %0, %0, %1")
DEF(dot4_i32_i8, "v_dot4_i32_i8 %0, %0, %1, %2")
DEF(msad_u8, "v_msad_u8 %0, %0, %1, %2")
DEF(sad_u32, "v_sad_u32 %0, %0, %1, %2")
DEF(min3_i16, "v_min3_i16 %0, %0, %1, %2")
DEF(med3_i16, "v_med3_i16 %0, %0, %1, %2")
int main() {
    uint32_t* d; hipMalloc(&d, 4 * 256 * 256 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    printf("CUs %d clock %d kHz\n", pr.multiProcessorCount, khz);
    const int blocks = 256 * 8; // 8 blocks of 4 waves per CU -> 8 waves per SIMD
#define RUN(name) { hipLaunchKernelGGL(k_##name, dim3(blocks), dim3(256), 0, 0, d, 1u); hipDeviceSynchronize(); \
        hipEventRecord(e0); hipLaunchKernelGGL(k_##name, dim3(blocks), dim3(256), 0, 0, d, 2u); hipEventRecord(e1); hipEventSynchronize(e1); \
        float ms; hipEventElapsedTime(&ms, e0, e1); double winstr = (double)blocks * 4 * ITER * 8; \
        double per_simd = winstr / (256.0 * 4); double cyc = ms * 1e-3 * (khz * 1e3) / per_simd; \
        printf("%-18s %8.3f ms  %6.2f cycles/wave-instr/SIMD (at nominal clock)  %7.2f Tlane-op/s\n", #name, ms, cyc, winstr * 64 / (ms * 1e-3) / 1e12); }
    RUN(add_u32) RUN(sub_u32) RUN(xor_b32) RUN(min_u32) RUN(med3_i32) RUN(min3_u32) RUN(sad_u16) RUN(sad_u8) RUN(bfe_u32) RUN(perm_b32)
    RUN(mad_i24) RUN(mul_i24) RUN(lshl_or) RUN(add3) RUN(xad) RUN(ashr) RUN(pk_add_i16) RUN(pk_add_i16_clamp) RUN(pk_sub_i16) RUN(pk_min_i16)
    RUN(pk_max_i16) RUN(pk_min_u16) RUN(pk_mad_i16) RUN(pk_mul_lo_u16) RUN(pk_ashr_i16) RUN(pk_lshl_b16) RUN(cndmask) RUN(cmp_eq) RUN(and_or) RUN(bfi)
    RUN(fma_f32) RUN(pk_fma_f16) RUN(pk_max_f16) RUN(mov_dpp)
    RUN(min_f32) RUN(max_f32) RUN(med3_f32) RUN(min3_f32) RUN(add_f32) RUN(add_f32_abs) RUN(sub_f32) RUN(mul_f32) RUN(cvt_f32_ubyte1) RUN(cvt_pk_u8_f32) RUN(cvt_u32_f32) RUN(cvt_f32_i32)
    RUN(cmp_eq_f32_cnd) RUN(cmp_eq_u32_cnd) RUN(cmp_sgpr_cnd) RUN(and_b32) RUN(or_b32) RUN(not_b32) RUN(min_i32) RUN(max_i32) RUN(lshlrev) RUN(lshrrev) RUN(sub_sdwa) RUN(and_sdwa) RUN(mov_b32) RUN(alignbit) RUN(fma_mix) RUN(subrev_f32_neg)
    RUN(sub_i32_clamp) RUN(add_i32_clamp) RUN(sub_u32_clamp) RUN(add_u32_e64) RUN(lshl_sdwa) RUN(med3_u32) RUN(max3_u32) RUN(add_i16_clamp) RUN(sub_co) RUN(subb_co) RUN(xor3) RUN(lshl_add) RUN(add_lshl) RUN(or3)
    RUN(pk_add_f32) RUN(pk_mul_f32) RUN(pk_fma_f32) RUN(pk_mov_b32) RUN(lshlrev_b64) RUN(cvt_f32_ubyte0) RUN(cvt_i32_f32) RUN(med3_f32_abs) RUN(min3_f32_abs) RUN(max3_f32)
    RUN(add_f32_clamp) RUN(fma_f32_clamp) RUN(mul_legacy) RUN(ldexp_f32) RUN(cvt_pk_i16_i32) RUN(cvt_pk_u16_u32) RUN(cvt_pknorm_i16) RUN(sat_pk_u8_i16) RUN(pk_sub_i16_clamp) RUN(pk_sub_u16_clamp) RUN(pk_add_u16_clamp) RUN(pk_max_u16) RUN(min_u16) RUN(dot4_i32_i8) RUN(msad_u8) RUN(sad_u32) RUN(min3_i16) RUN(med3_i16)
    return 0;
}
