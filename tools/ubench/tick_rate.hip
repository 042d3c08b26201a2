// Calibrates s_memtime (what __builtin_readcyclecounter reads on gfx950) against s_memrealtime (100 MHz) and wall clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint64_t* out, int n)
{
    uint64_t r0, r1;
    asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(r0));
    uint64_t t0 = __builtin_readcyclecounter();
    uint32_t a = threadIdx.x;
    for (int i = 0; i < n; i++) asm volatile("v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1\n v_add_u32 %0, %0, 1" : "+v"(a));
    uint64_t t1 = __builtin_readcyclecounter();
    asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(r1));
    if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = a; }
}
int main()
{
    uint64_t* d; (void)hipMalloc(&d, 64); uint64_t h[3];
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int n : {1000000, 4000000}) {
        (void)hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, n); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); (void)hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("n=%d: %.3f ms wall, s_memtime delta %llu (%.1f MHz), s_memrealtime delta %llu (%.1f MHz), %.2f s_memtime ticks per v_add, %.3f ns per v_add\n", n, ms,
               (unsigned long long)h[0], h[0] / (ms * 1e3), (unsigned long long)h[1], h[1] / (ms * 1e3), (double)h[0] / (4.0 * n), ms * 1e6 / (4.0 * n));
    }
    return 0;
}
