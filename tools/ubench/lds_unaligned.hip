// Micro-benchmark (gfx950): do 16- and 32-bit LDS accesses work at addresses that are not multiples of their size, and at what rate?
// (The question behind a check node that keeps TWO ROWS of one entry in the halves of a register: rows 2j, 2j + 1 of an entry with an odd
// rotation lie at an odd byte address.) Every lane reads / writes at base + stride * lane + off, off = 0 .. 3; results are checked against
// the byte pattern, rates are cycles per wave-instruction and CU with 12 waves resident.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#define ITER 2048
template <int KIND /*0: u8 x2, 1: u16, 2: b32, 3: write b16, 4: write b8 x2*/>
__global__ __launch_bounds__(768) void k(uint32_t* out, int off, unsigned long long* cyc)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[8192];
    for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (uint8_t)(i * 7 + 3);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t addr = (uint32_t)(size_t)lds + (uint32_t)(wave * 512 + (KIND == 2 ? 4 : 2) * lane + off);
    uint32_t acc = 0, v = 0, w = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < ITER; i++) {
        if (KIND == 0) { asm volatile("ds_read_u8 %0, %2\n\tds_read_u8 %1, %2 offset:1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v), "=v"(w) : "v"(addr) : "memory"); v |= w << 8; acc += v; }
        if (KIND == 1) { asm volatile("ds_read_u16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory"); acc += v; }
        if (KIND == 2) { asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory"); acc += v; }
        if (KIND == 3) { v = acc + i; asm volatile("ds_write_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" :: "v"(addr), "v"(v) : "memory"); acc += 0x0101; }
        if (KIND == 4) { v = acc + i; w = v >> 8; asm volatile("ds_write_b8 %0, %1\n\tds_write_b8 %0, %2 offset:1\n\ts_waitcnt lgkmcnt(0)" :: "v"(addr), "v"(v), "v"(w) : "memory"); acc += 0x0101; }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (KIND >= 3) { __syncthreads(); acc = lds[wave * 512 + 2 * lane + off] | (lds[wave * 512 + 2 * lane + off + 1] << 8); }
    out[blockIdx.x * blockDim.x + threadIdx.x] = KIND < 3 ? v : acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int KIND> void run(const char* name)
{
    uint32_t* d; unsigned long long* c; hipMalloc(&d, 768 * 512 * 4); hipMalloc(&c, 8);
    for (int off = 0; off < 4; off++) {
        hipLaunchKernelGGL(k<KIND>, dim3(512), dim3(768), 0, 0, d, off, c);
        std::vector<uint32_t> h(768); unsigned long long cy = 0;
        hipMemcpy(h.data(), d, 768 * 4, hipMemcpyDeviceToHost); hipMemcpy(&cy, c, 8, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int t = 0; t < 768 && KIND < 3; t++) {
            const int a = (t >> 6) * 512 + (KIND == 2 ? 4 : 2) * (t & 63) + off;
            uint32_t want = 0;
            for (int b = 0; b < (KIND == 2 ? 4 : 2); b++) want |= (uint32_t)(uint8_t)((a + b) * 7 + 3) << (8 * b);
            bad += h[t] != want;
        }
        for (int t = 0; t < 768 && KIND >= 3; t++) bad += h[t] != ((0x0101u * (ITER - 1) + (ITER - 1)) & 0xffffu); // what the last iteration wrote, read back bytewise
        printf("%-14s off %d: %s  %.1f cycles per wave-instruction and CU (12 waves, %d iterations%s)\n", name, off, bad ? "WRONG" : "ok",
               (double)cy / ITER / 12.0 / ((KIND == 0 || KIND == 4) ? 2 : 1), ITER, (KIND == 0 || KIND == 4) ? ", two instructions" : "");
    }
    hipFree(d); hipFree(c);
}
int main()
{
    run<0>("2 x ds_read_u8"); run<1>("ds_read_u16"); run<2>("ds_read_b32"); run<3>("ds_write_b16"); run<4>("2 x ds_write_b8");
    return 0;
}
