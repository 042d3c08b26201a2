// Does SALU / LDS issue steal VALU issue slots on gfx950?  8 independent VALU per trip + K scalar ops per trip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define ITER 8192
template <int K, int VOP>
__global__ __launch_bounds__(768) void k_mix(uint32_t* out, uint32_t seed)
{
    uint32_t a0 = threadIdx.x + seed, a1 = a0 * 3, a2 = a0 * 5, a3 = a0 * 7, a4 = a0 * 11, a5 = a0 * 13, a6 = a0 * 17, a7 = a0 * 19;
    uint32_t b = seed * 77 + 5;
    uint32_t s0 = seed, s1 = seed + 1, s2 = seed + 2, s3 = seed + 3;
    for (int i = 0; i < ITER; i++) {
        if (VOP == 0) {
            asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                         "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        } else {
            asm volatile("v_med3_i32 %0, %0, %8, %8\n v_med3_i32 %1, %1, %8, %8\n v_med3_i32 %2, %2, %8, %8\n v_med3_i32 %3, %3, %8, %8\n"
                         "v_med3_i32 %4, %4, %8, %8\n v_med3_i32 %5, %5, %8, %8\n v_med3_i32 %6, %6, %8, %8\n v_med3_i32 %7, %7, %8, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        }
        if (K >= 4) asm volatile("s_add_u32 %0, %0, 1\n s_add_u32 %1, %1, 3\n s_xor_b32 %2, %2, %0\n s_lshl_b32 %3, %3, 1\n" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        if (K >= 8) asm volatile("s_add_u32 %0, %0, 5\n s_add_u32 %1, %1, 7\n s_xor_b32 %2, %2, %1\n s_lshr_b32 %3, %3, 1\n" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
        if (K >= 16) asm volatile("s_add_u32 %0, %0, 5\n s_add_u32 %1, %1, 7\n s_xor_b32 %2, %2, %1\n s_lshr_b32 %3, %3, 1\n"
                                  "s_add_u32 %0, %0, 9\n s_add_u32 %1, %1, 11\n s_xor_b32 %2, %2, %0\n s_lshl_b32 %3, %3, 2\n" : "+s"(s0), "+s"(s1), "+s"(s2), "+s"(s3) : : "scc");
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ s0 ^ s1 ^ s2 ^ s3;
}
int main()
{
    uint32_t* d; hipMalloc(&d, 4 * 256 * 256 * 16);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int wpb : {8, 3}) {   // waves per SIMD: 8 (blocks of 256 thr x 8 per CU) or 3 (768-thread block per CU)
        const int threads = wpb == 8 ? 256 : 768, blocks = wpb == 8 ? 256 * 8 : 256;
#define RUN(K, V) { hipLaunchKernelGGL((k_mix<K, V>), dim3(blocks), dim3(threads), 0, 0, d, 1u); hipDeviceSynchronize(); \
        { hipError_t le = hipGetLastError(); if (le != hipSuccess) printf("launch error %s\n", hipGetErrorString(le)); } hipEventRecord(e0); for (int r = 0; r < 3; r++) hipLaunchKernelGGL((k_mix<K, V>), dim3(blocks), dim3(threads), 0, 0, d, 2u); hipEventRecord(e1); hipEventSynchronize(e1); \
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3; double trips = (double)ITER * wpb; \
        printf("waves/SIMD %d  %s x8 + %2d SALU per trip: %7.3f ms  -> %6.2f cycles(2.4GHz) per trip per SIMD-wave-slot\n", wpb, V ? "med3" : "add ", K, ms, ms * 1e-3 * 2.4e9 / trips); }
        RUN(0, 0) RUN(4, 0) RUN(8, 0) RUN(16, 0) RUN(0, 1) RUN(4, 1) RUN(8, 1) RUN(16, 1)
    }
    return 0;
}
