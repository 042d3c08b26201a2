// Micro-benchmark (gfx950): what a wave costs when it is (nearly) alone on its SIMD -- the regime of the ordered
// hazard steps of the LDPC sweep. Cycle stamps (s_memtime) around straight-line sequences, one workgroup per CU,
// W waves per SIMD. Reports cycles per instruction for: a dependent VALU chain, 4-way independent VALU, VALU with
// EXEC = 0, LDS write->read round trip, ds_bpermute, DPP, and a scalar load from a warm line.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 64
#define STR2(x) #x
#define STR(x) STR2(x)
#define REP8(x) x x x x x x x x
#define REP64(x) REP8(REP8(x))

template <int MODE>
__global__ __launch_bounds__(1024) void k(uint64_t* out, const uint32_t* tab, uint32_t seed)
{
    __shared__ uint32_t sm[2048];
    uint32_t a = threadIdx.x + seed, b = a * 3 + 1, c = a * 5 + 2, d = a * 7 + 3, e = seed + 11;
    sm[threadIdx.x] = a; sm[threadIdx.x + 1024] = b;
    __syncthreads();
    uint32_t addr = (threadIdx.x & 1023) * 4;
    uint64_t t0 = 0, t1 = 0;
    for (int rep = 0; rep < 3; rep++) { // last repetition counts (warm instruction cache)
        t0 = __builtin_readcyclecounter();
        if (MODE == 0) { asm volatile(REP64("v_add_u32 %0, %0, %1\n") : "+v"(a) : "v"(e)); }
        if (MODE == 1) { asm volatile(REP64("v_med3_i32 %0, %0, %1, %1\n") : "+v"(a) : "v"(e)); }
        if (MODE == 2) { asm volatile(REP8(REP8("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n")) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e)); }
        if (MODE == 3) { asm volatile(REP8(REP8("v_med3_i32 %0, %0, %4, %4\n v_med3_i32 %1, %1, %4, %4\n v_med3_i32 %2, %2, %4, %4\n v_med3_i32 %3, %3, %4, %4\n")) : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e)); }
        if (MODE == 4) { // EXEC = 0
            asm volatile("s_mov_b64 s[20:21], exec\n s_mov_b64 exec, 0\n" REP64("v_add_u32 %0, %0, %1\n") "s_mov_b64 exec, s[20:21]\n" : "+v"(a) : "v"(e) : "s20", "s21");
        }
        if (MODE == 5) { // LDS write -> read of the same address, dependent
            asm volatile(REP8("ds_write_b32 %1, %0\n ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_add_u32 %0, %0, 1\n") : "+v"(a) : "v"(addr) : "memory");
        }
        if (MODE == 6) { // LDS read only, dependent address
            asm volatile(REP8("ds_read_b32 %0, %1\n s_waitcnt lgkmcnt(0)\n v_and_b32 %0, 0xffc, %0\n v_add_u32 %1, %1, %0\n v_and_b32 %1, 0xffc, %1\n") : "+v"(a), "+v"(addr) : : "memory");
        }
        if (MODE == 7) { // ds_bpermute dependent
            asm volatile(REP8("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)\n") : "+v"(a) : "v"(addr) : "memory");
        }
        if (MODE == 8) { // DPP row_shr dependent
            asm volatile(REP64("v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n") : "+v"(a));
        }
        if (MODE == 9) { // scalar load, dependent
            const uint32_t* p = tab;
            uint32_t s = 0;
            asm volatile(REP8("s_load_dword %0, %1, %0\n s_waitcnt lgkmcnt(0)\n s_and_b32 %0, %0, 0\n") : "+s"(s) : "s"(p) : "scc", "memory");
            a += s;
        }
        if (MODE == 10) { // mixed: dependent chain of add (full-rate) alternating with med3
            asm volatile(REP8(REP8("v_add_u32 %0, %0, %1\n v_med3_i32 %0, %0, %1, %1\n")) : "+v"(a) : "v"(e));
        }
        if (MODE == 11) { // v_readlane + v_writelane style hand-over: readlane into SGPR then use as operand
            asm volatile(REP8(REP8("v_readlane_b32 s20, %0, 3\n v_add_u32 %0, s20, %0\n")) : "+v"(a) : : "s20");
        }
        if (MODE == 12) { // LDS byte write -> byte read, different addresses (in-order within a wave), dependent through data
            asm volatile(REP8("ds_write_b8 %1, %0\n ds_read_u8 %0, %1 offset:0\n s_waitcnt lgkmcnt(0)\n") : "+v"(a) : "v"(addr) : "memory");
        }
        t1 = __builtin_readcyclecounter();
    }
    if ((threadIdx.x & 63) == 0) { out[(blockIdx.x * 16 + (threadIdx.x >> 6)) * 2] = t1 - t0; out[8192 + (blockIdx.x * 16 + (threadIdx.x >> 6)) * 2] = t0; out[8192 + (blockIdx.x * 16 + (threadIdx.x >> 6)) * 2 + 1] = t1; }
    if (a == 0x12345678 && b == c && d == e) out[1] = a;
}

int main()
{
    uint64_t* d; hipMalloc(&d, 2 * 8 * 2 * 16 * 256);
    uint32_t* tab; hipMalloc(&tab, 4096); hipMemset(tab, 0, 4096);
    static uint64_t h[2 * 2 * 16 * 256];
    const char* names[] = {"dep v_add x64", "dep v_med3 x64", "indep4 v_add x256", "indep4 v_med3 x256", "EXEC=0 v_add x64", "lds wr->rd +add x8", "lds rd dep x8 (+3 valu)", "ds_bpermute dep x8",
                           "dpp row_shr dep x64", "s_load dep x8", "dep add,med3 x128", "readlane+add x128", "lds wr8->rd8 x8"};
    const int counts[] = {64, 64, 256, 256, 64, 8, 8, 8, 64, 8, 128, 128, 8};
    for (int waves : {1, 4, 8, 12}) {
        printf("---- %d wave(s) per workgroup, one workgroup per CU ----\n", waves);
#define RUN(M) { hipMemset(d, 0, sizeof(h)); hipLaunchKernelGGL((k<M>), dim3(256), dim3(64 * waves), 0, 0, d, tab, 1u); hipDeviceSynchronize(); \
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); double s = 0; int n = 0; for (int b = 0; b < 256; b++) for (int w = 0; w < waves; w++) { s += (double)h[(b * 16 + w) * 2]; n++; } \
        double span = 0; for (int b = 0; b < 256; b++) { uint64_t lo = ~0ull, hi = 0; for (int w = 0; w < waves; w++) { uint64_t a0 = h[8192 + (b * 16 + w) * 2], a1 = h[8192 + (b * 16 + w) * 2 + 1]; if (a0 < lo) lo = a0; if (a1 > hi) hi = a1; } span += (double)(hi - lo); } \
        printf("  %-26s %9.1f cycles per wave  %7.2f per instr/iteration   workgroup span %9.1f\n", names[M], s / n, s / n / counts[M], span / 256); }
        RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7) RUN(8) RUN(9) RUN(10) RUN(11) RUN(12)
    }
    return 0;
}
