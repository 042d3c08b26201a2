// Micro-benchmark (gfx950): where do the waves of two 6-wave workgroups that share a CU land? Each wave records HW_ID
// (s_getreg_b32 hwreg(HW_REG_HW_ID)): wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh[12] se[15:13] ... and XCC_ID.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <map>
#include <vector>
#include <chrono>
#include <string>
__global__ void k(uint32_t* out, int spin)
{
    extern __shared__ uint8_t sm[];
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    sm[threadIdx.x] = (uint8_t)hw;
    uint64_t t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (uint64_t)spin) { }
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
}
// the same with the occupancy capped at three waves per SIMD (the compiler pads the kernel's VGPR allocation to 136 .. 168): do two
// 6-wave workgroups then land 3,3,3,3 -- is the dispatcher's placement resource-aware, or does the second workgroup not fit at all?
// (amdgpu_waves_per_eu(3, 3) does NOT pad the allocation with this compiler: .amdhsa_next_free_vgpr stays at what the code uses; a
// clobbered high register does)
__global__ void k3(uint32_t* out, int spin)
{
    extern __shared__ uint8_t sm[];
    asm volatile("v_mov_b32 v140, 0" ::: "v140");
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    sm[threadIdx.x] = (uint8_t)hw;
    uint64_t t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (uint64_t)spin) { }
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2] = hw; out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + 1] = xcc; }
}
int main(int argc, char** argv)
{
    const bool cap3 = argc > 3 && atoi(argv[3]) != 0;
    const int waves = argc > 1 ? atoi(argv[1]) : 6, lds = argc > 2 ? atoi(argv[2]) : 75000, blocks = 512;
    uint32_t* d; (void)hipMalloc(&d, blocks * 16 * 2 * 4); (void)hipMemset(d, 0xff, blocks * 16 * 2 * 4);
    (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    (void)hipFuncSetAttribute((const void*)k3, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    auto t0 = std::chrono::steady_clock::now();
    if (cap3) hipLaunchKernelGGL(k3, dim3(blocks), dim3(64 * waves), lds, 0, d, 2000000);
    else hipLaunchKernelGGL(k, dim3(blocks), dim3(64 * waves), lds, 0, d, 2000000);
    (void)hipDeviceSynchronize();
    printf("%s: %.2f ms for %d workgroups spinning 2 M ticks (20 ms at 100 MHz) each\n", cap3 ? "capped at 3 waves per SIMD" : "plain",
           std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count(), blocks);
    std::vector<uint32_t> h(blocks * 16 * 2); (void)hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    std::map<uint32_t, std::vector<int>> per_cu; // key: xcc, se, sh, cu -> simd ids of resident waves with block id
    for (int b = 0; b < blocks; b++) for (int w = 0; w < waves; w++) {
        const uint32_t hw = h[(b * 16 + w) * 2], xcc = h[(b * 16 + w) * 2 + 1] & 0xf;
        const uint32_t key = (xcc << 16) | (((hw >> 13) & 7) << 8) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf);
        per_cu[key].push_back(b * 100 + ((hw >> 4) & 3));
    }
    int shown = 0; std::map<std::string, int> patterns;
    for (auto& kv : per_cu) {
        int cnt[4] = {0, 0, 0, 0};
        std::map<int, std::vector<int>> byblock;
        for (int v : kv.second) { cnt[v % 100]++; byblock[v / 100].push_back(v % 100); }
        char buf[64]; snprintf(buf, sizeof buf, "%d,%d,%d,%d (%zu workgroups)", cnt[0], cnt[1], cnt[2], cnt[3], byblock.size());
        patterns[buf]++;
        if (shown++ < 4) { printf("CU %06x:", kv.first); for (auto& bb : byblock) { printf("  wg%d simds", bb.first); for (int s : bb.second) printf(" %d", s); } printf("\n"); }
    }
    printf("waves per SIMD over the first residency (%d-wave workgroups, %d B LDS each), %zu CUs seen:\n", waves, lds, per_cu.size());
    for (auto& p : patterns) printf("  %s : %d CUs\n", p.first.c_str(), p.second);
    return 0;
}
