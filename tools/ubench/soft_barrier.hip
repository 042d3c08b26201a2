// Micro-benchmark (gfx950): what a barrier of SIX of a workgroup's twelve waves costs -- the per-frame software barrier of the
// LDPC sweep kernels (ldpc_kernel.hpp, frame_barrier) against the hardware barrier of all twelve, and whether S_WAKEUP (the ISA's
// "fBarrier speedup": a wave pings the sleeping waves of its workgroup) shortens it.
// One workgroup of 12 waves per CU, two halves of six waves; every wave runs REPS episodes of { W dependent VALU instructions; barrier }.
// The waves of a half arrive together (same work), so an episode costs W x ~4.5 cycles + the barrier's own latency.
//   mode 0  s_waitcnt lgkmcnt(0) + s_barrier (all twelve waves)
//   mode 1  LDS counter: ds_add (lane 0), poll with s_sleep 1                      (the kernel's frame_barrier)
//   mode 2  the same without s_sleep
//   mode 3  ds_add, s_wakeup; waiters poll with s_sleep 8
//   mode 4  ds_add_rtn: the LAST arriver (old value == expected - 1) bumps a release word and pings; waiters poll the release word, s_sleep 8
//   mode 5  as 1 with s_sleep 0 ... (s_sleep 0 = yield)
// Second part: a hand-over chain inside a half (wave w waits for wave w - 1's flag, then sets its own): cycles per hop, hardware barrier
// per hop against flag polling (with / without s_sleep, with s_wakeup).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define REPS 256

typedef __attribute__((address_space(3))) int lds_i32_t;

template <int MODE, int W>
__global__ __launch_bounds__(768) void kbar(uint64_t* out, uint32_t seed)
{
    __shared__ int ctr[64];
    const int half = threadIdx.x >= 384 ? 1 : 0;
    const int lane = threadIdx.x & 63;
    volatile lds_i32_t* c = (volatile lds_i32_t*)&ctr[16 * half];
    volatile lds_i32_t* rel = (volatile lds_i32_t*)&ctr[16 * half + 8];
    if (threadIdx.x < 64) ctr[threadIdx.x] = 0;
    __syncthreads();
    uint32_t a = threadIdx.x + seed;
    int epoch = 0, gen = 0;
    uint64_t t0 = 0, t1 = 0;
    for (int rep = 0; rep < 2; rep++) {
        t0 = __builtin_readcyclecounter();
        for (int r = 0; r < REPS; r++) {
#pragma unroll
            for (int w = 0; w < W; w++) asm volatile("v_add_u32 %0, %0, 1" : "+v"(a));
            if (MODE == 0) { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
            else if (MODE == 4) {
                gen++;
                int old = 0;
                if (lane == 0) old = __hip_atomic_fetch_add(const_cast<lds_i32_t*>(c), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                old = __builtin_amdgcn_readfirstlane(old);
                if (old == 6 * gen - 1) { if (lane == 0) *rel = gen; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_wakeup" ::: "memory"); }
                else while (*rel - gen < 0) __builtin_amdgcn_s_sleep(8);
            } else {
                epoch += 6;
                asm volatile("" ::: "memory");
                if (lane == 0) __hip_atomic_fetch_add(const_cast<lds_i32_t*>(c), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (MODE == 3) asm volatile("s_wakeup" ::: "memory");
                if (MODE == 1) while (*c - epoch < 0) __builtin_amdgcn_s_sleep(1);
                if (MODE == 2) while (*c - epoch < 0) { }
                if (MODE == 3) while (*c - epoch < 0) __builtin_amdgcn_s_sleep(8);
                if (MODE == 5) while (*c - epoch < 0) __builtin_amdgcn_s_sleep(0);
                asm volatile("" ::: "memory");
            }
        }
        t1 = __builtin_readcyclecounter();
    }
    if (lane == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    if (a == 0x12345678) out[0] = a;
}

// hand-over chain: in every episode wave 0 of the half starts, wave w runs after wave w - 1 (six hops), then all meet at a hardware barrier
template <int MODE>
__global__ __launch_bounds__(768) void kchain(uint64_t* out, uint32_t seed)
{
    __shared__ int flag[64];
    const int half = threadIdx.x >= 384 ? 1 : 0;
    const int lane = threadIdx.x & 63, wv = (threadIdx.x - 384 * half) >> 6;
    volatile lds_i32_t* f = (volatile lds_i32_t*)&flag[16 * half];
    if (threadIdx.x < 64) flag[threadIdx.x] = 0;
    __syncthreads();
    uint32_t a = threadIdx.x + seed;
    uint64_t t0 = 0, t1 = 0;
    for (int rep = 0; rep < 2; rep++) {
        t0 = __builtin_readcyclecounter();
        for (int r = 1; r <= REPS; r++) {
            if (MODE == 0) { // hardware: six barrier-separated steps
                for (int s = 0; s < 6; s++) {
                    if (s == wv) asm volatile("v_add_u32 %0, %0, 1\n\tv_add_u32 %0, %0, 1\n\tv_add_u32 %0, %0, 1\n\tv_add_u32 %0, %0, 1" : "+v"(a));
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                }
            } else {
                const int want = 6 * (rep * REPS + r - 1) + wv; // flag value that lets wave wv run in episode r
                if (MODE == 1) while (*f - want < 0) __builtin_amdgcn_s_sleep(1);
                if (MODE == 2) while (*f - want < 0) { }
                if (MODE == 3) while (*f - want < 0) __builtin_amdgcn_s_sleep(8);
                asm volatile("v_add_u32 %0, %0, 1\n\tv_add_u32 %0, %0, 1\n\tv_add_u32 %0, %0, 1\n\tv_add_u32 %0, %0, 1" : "+v"(a));
                if (lane == 0) *f = want + 1;
                if (MODE == 3) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_wakeup" ::: "memory");
                // (the next episode's wave 0 waits for 6 r: the last wave of this episode)
            }
        }
        t1 = __builtin_readcyclecounter();
    }
    if (lane == 0) out[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
    if (a == 0x12345678) out[0] = a;
}

int main()
{
    uint64_t* d; hipMalloc(&d, 8 * 16 * 256);
    static uint64_t h[16 * 256];
#define RUN(K, name, per) { hipMemset(d, 0, sizeof(h)); hipLaunchKernelGGL(K, dim3(256), dim3(768), 0, 0, d, 1u); if (hipDeviceSynchronize() != hipSuccess) { printf("%s: launch failed\n", name); return 1; } \
        hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost); double s = 0; int n = 0; for (int b = 0; b < 256; b++) for (int w = 0; w < 12; w++) { s += (double)h[b * 16 + w]; n++; } \
        printf("  %-64s %9.1f cycles per %s\n", name, s / n / REPS / (per), (per) == 1 ? "episode" : "hop"); }
    printf("---- barrier episodes: W dependent v_add, then a barrier of the half (12 waves per CU, 256 workgroups) ----\n");
    RUN((kbar<0, 8>),  "W=8   hardware s_barrier (twelve waves)", 1)
    RUN((kbar<1, 8>),  "W=8   LDS counter, poll + s_sleep 1 (kernel)", 1)
    RUN((kbar<2, 8>),  "W=8   LDS counter, poll without sleep", 1)
    RUN((kbar<5, 8>),  "W=8   LDS counter, poll + s_sleep 0", 1)
    RUN((kbar<3, 8>),  "W=8   LDS counter + s_wakeup, poll + s_sleep 8", 1)
    RUN((kbar<4, 8>),  "W=8   ds_add_rtn, last arriver releases + s_wakeup, s_sleep 8", 1)
    RUN((kbar<0, 64>), "W=64  hardware s_barrier (twelve waves)", 1)
    RUN((kbar<1, 64>), "W=64  LDS counter, poll + s_sleep 1 (kernel)", 1)
    RUN((kbar<2, 64>), "W=64  LDS counter, poll without sleep", 1)
    RUN((kbar<3, 64>), "W=64  LDS counter + s_wakeup, poll + s_sleep 8", 1)
    RUN((kbar<4, 64>), "W=64  ds_add_rtn, last arriver releases + s_wakeup, s_sleep 8", 1)
    printf("---- hand-over chain: wave w runs after wave w - 1 (4 v_add each), six hops per episode ----\n");
    RUN((kchain<0>), "hardware barrier per hop", 6)
    RUN((kchain<1>), "flag, poll + s_sleep 1", 6)
    RUN((kchain<2>), "flag, poll without sleep", 6)
    RUN((kchain<3>), "flag + s_wakeup, poll + s_sleep 8", 6)
    return 0;
}
