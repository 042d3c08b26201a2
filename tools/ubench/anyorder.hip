// tools/ubench/anyorder.hip -- does hipExtAnyOrderLaunch clear the AQL barrier bit on gfx950, i.e. do two kernels of ONE stream overlap?
// Each kernel: `wgs` workgroups spinning for `ticks` s_memtime ticks. Serial: total = 2 x one; overlapped (wgs <= half the CUs): total = one.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <chrono>
__global__ void spin(long long ticks, int* sink) {
    long long t0 = __builtin_readcyclecounter();
    int x = 0;
    while (__builtin_readcyclecounter() - t0 < ticks) x++;
    if (x == -1) *sink = x;
}
static double run(int flags, bool two_streams, int wgs, long long ticks) {
    hipStream_t s0, s1; hipStreamCreate(&s0); hipStreamCreate(&s1);
    int* sink; hipMalloc(&sink, 4);
    void* args[] = {&ticks, &sink};
    for (int w = 0; w < 2; ++w) {  // warm, then timed
        hipDeviceSynchronize();
        auto t = std::chrono::steady_clock::now();
        hipExtLaunchKernel((const void*)spin, dim3(wgs), dim3(64), args, 0, s0, nullptr, nullptr, flags);
        hipExtLaunchKernel((const void*)spin, dim3(wgs), dim3(64), args, 0, two_streams ? s1 : s0, nullptr, nullptr, flags);
        hipDeviceSynchronize();
        double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
        if (w == 1) { hipFree(sink); hipStreamDestroy(s0); hipStreamDestroy(s1); return ms; }
    }
    return 0;
}
int main() {
    long long ticks = 100000000LL / 10;  // 100 MHz counter: 0.1 s
    printf("one stream, flags 0:            %.1f ms\n", run(0, false, 64, ticks));
    printf("one stream, hipExtAnyOrderLaunch: %.1f ms\n", run(hipExtAnyOrderLaunch, false, 64, ticks));
    printf("two streams, flags 0:           %.1f ms\n", run(0, true, 64, ticks));
    return 0;
}
