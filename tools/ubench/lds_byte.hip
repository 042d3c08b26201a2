// Micro-benchmark (gfx950): throughput of the LDS byte accesses of the LDPC sweep. 12 waves per CU; each wave issues streams of
// ds_read_u8 / ds_write_b8 (64 consecutive bytes per instruction, arbitrary alignment, like a rotated 360-byte window), and for
// comparison ds_read_b32 / ds_write_b32 (64 consecutive dwords) and 16-bit accesses. Reports cycles per wave-instruction per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP8(x) x x x x x x x x
template <int MODE>
__global__ __launch_bounds__(768) void k(uint64_t* out, uint32_t seed)
{
    extern __shared__ uint8_t sm[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 150000 / 4; i += 768) reinterpret_cast<uint32_t*>(sm)[i] = i * seed;
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63;
    // byte modes: lane l touches byte base + l; base differs per wave and per step (odd offsets)
    uint32_t a0 = (uint32_t)(wave * 10007 + 3 + lane) % 140000u, a1 = a0 + 361, a2 = a0 + 2 * 361 + 1, a3 = a0 + 3 * 361 + 2;
    uint32_t d0 = a0 * 4 % 140000u & ~3u;
    uint32_t v0 = seed, v1 = seed + 1, v2 = seed + 2, v3 = seed + 3, acc = 0;
    uint64_t t0 = 0, t1 = 0;
    for (int rep = 0; rep < 2; rep++) {
        __syncthreads();
        t0 = __builtin_readcyclecounter();
        for (int it = 0; it < 64; it++) {
            if (MODE == 0) asm volatile(REP8("ds_read_u8 %0, %4\n ds_read_u8 %1, %5\n ds_read_u8 %2, %6\n ds_read_u8 %3, %7\n") "s_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
            if (MODE == 1) asm volatile(REP8("ds_write_b8 %4, %0\n ds_write_b8 %5, %1\n ds_write_b8 %6, %2\n ds_write_b8 %7, %3\n") "s_waitcnt lgkmcnt(0)" : : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
            if (MODE == 2) asm volatile(REP8("ds_read_b32 %0, %4\n ds_read_b32 %1, %4 offset:256\n ds_read_b32 %2, %4 offset:512\n ds_read_b32 %3, %4 offset:768\n") "s_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(d0) : "memory");
            if (MODE == 3) asm volatile(REP8("ds_write_b32 %4, %0\n ds_write_b32 %4, %1 offset:256\n ds_write_b32 %4, %2 offset:512\n ds_write_b32 %4, %3 offset:768\n") "s_waitcnt lgkmcnt(0)" : : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(d0) : "memory");
            if (MODE == 4) asm volatile(REP8("ds_read_u8 %0, %4\n ds_write_b8 %5, %1\n ds_read_u8 %2, %6\n ds_write_b8 %7, %3\n") "s_waitcnt lgkmcnt(0)" : "=&v"(v0), "+v"(v1), "=&v"(v2), "+v"(v3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "memory");
            if (MODE == 5) { // 16-bit accesses at even addresses: 64 lanes x 2 bytes
                uint32_t e0 = (a0 * 2) % 140000u & ~1u;
                asm volatile(REP8("ds_read_u16 %0, %4\n ds_read_u16 %1, %4 offset:128\n ds_read_u16 %2, %4 offset:256\n ds_read_u16 %3, %4 offset:384\n") "s_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(e0) : "memory");
            }
            if (MODE == 6) { // byte reads with a 4-byte lane stride (one byte per dword: no two lanes share a dword)
                uint32_t s0 = (uint32_t)(wave * 10007 + 3 + lane * 4) % 140000u;
                asm volatile(REP8("ds_read_u8 %0, %4\n ds_read_u8 %1, %4 offset:1\n ds_read_u8 %2, %4 offset:2\n ds_read_u8 %3, %4 offset:3\n") "s_waitcnt lgkmcnt(0)" : "=&v"(v0), "=&v"(v1), "=&v"(v2), "=&v"(v3) : "v"(s0) : "memory");
            }
            if (MODE == 7) { // byte writes with a 4-byte lane stride
                uint32_t s0 = (uint32_t)(wave * 10007 + 3 + lane * 4) % 140000u;
                asm volatile(REP8("ds_write_b8 %4, %0\n ds_write_b8 %4, %1 offset:1\n ds_write_b8 %4, %2 offset:2\n ds_write_b8 %4, %3 offset:3\n") "s_waitcnt lgkmcnt(0)" : : "v"(v0), "v"(v1), "v"(v2), "v"(v3), "v"(s0) : "memory");
            }
            acc += v0 ^ v1 ^ v2 ^ v3;
        }
        __syncthreads();
        t1 = __builtin_readcyclecounter();
    }
    if (tid == 0) { out[blockIdx.x * 2] = t1 - t0; out[blockIdx.x * 2 + 1] = acc; }
}
int main()
{
    uint64_t* d; (void)hipMalloc(&d, 16 * 256); uint64_t h[512];
    const char* names[] = {"ds_read_u8 consecutive bytes", "ds_write_b8 consecutive bytes", "ds_read_b32 consecutive dwords", "ds_write_b32 consecutive dwords",
                           "read_u8 + write_b8 mixed", "ds_read_u16 consecutive", "ds_read_u8 stride 4", "ds_write_b8 stride 4"};
#define RUN(M) { (void)hipFuncSetAttribute((const void*)k<M>, hipFuncAttributeMaxDynamicSharedMemorySize, 150000); hipLaunchKernelGGL((k<M>), dim3(256), dim3(768), 150000, 0, d, 3u); (void)hipDeviceSynchronize(); \
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); double s = 0; for (int b = 0; b < 256; b++) s += (double)h[2 * b]; s /= 256; \
    printf("%-34s %9.0f cycles for %d wave-instr per CU -> %6.2f cycles per wave-instruction per CU\n", names[M], s, 12 * 64 * 32, s / (12 * 64 * 32)); }
    RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6) RUN(7)
    return 0;
}
