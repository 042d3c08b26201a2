// ldpc_kernel.hpp -- device code of the layered LDPC decoder (included by ldpc_hip.hip for the constants and by
// the per-variant translation units ldpc_inst_*.hip, which instantiate one DMAX each so that the seven kernel
// variants compile in parallel).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include "demap_math.hpp"

#ifndef DVBS2_FWALK_MAXDEG
#define DVBS2_FWALK_MAXDEG 12 // measured (round 4, interleaved A/B): 8 -> B2 +8 %, B4 +0.5 %; 12 -> 3/5 normal +1 %, T2 2/3 +4 %, the one-frame class-12 tables +5 %; 16 -> B7 -1 %; 28 -> 8/9 normal -8 %, 5/6 -3 %
#endif
#ifndef DVBS2_FWALK
#define DVBS2_FWALK 1 // float walk with four rows in flight in the plain lane chain of the degree class 8 (check_node_hazard)
#endif

// Lane-chain layers (check_node_hazard), round 4: the pair's LLR bytes read with the regular entries, and the float walk on absolute LDS
// addresses. Largest check degree that gets them -- measured (interleaved A/B, whole tables): degree <= 8: B4 +0.7 %, 9/20 ... S2_TABLE_B3 +1.8 %,
// B1 +1 %, B2 -0.6 %; degree class 12: 3/5 normal 0, 2/3 normal and T2 2/3 -4 % (register allocation of their one-frame builds) -- and the
// degree-5..8 instantiations INSIDE the class-12 kernel cost its tables 3-7 % as well (S2X 99/180 ... S2X_TABLE_B4 -7 %), so the switch is the
// kernel's class (CLASS8 = DMAX <= 8), not the check degree
#ifndef DVBS2_EARLY_PAIR_MAXDEG
#define DVBS2_EARLY_PAIR_MAXDEG 8
#endif
#ifndef DVBS2_WALK_ABS_MAXDEG
#define DVBS2_WALK_ABS_MAXDEG 8
#endif
// The new messages of a layer are stored behind an explicit `s_waitcnt vmcnt(0)`. The compiler waits for the NEXT layer's prefetched
// messages (loaded at the head of this layer) with vmcnt(0) -- loads and stores share the counter and across this loop's control flow it
// cannot count them -- and, left alone, places that wait at their first use, i.e. right AFTER the stores it has just issued: every wave of the
// workgroup then sat through the L2's acknowledgement of its stores at every layer boundary. Waiting first costs nothing (the prefetch
// is a layer old) and leaves the stores a whole layer to complete; no register is added. Measured (interleaved A/B): 1/4 normal +7 %,
// B4 +3.6 %, 1/3 normal +8 %, 3/4 normal +3 % (23 tables, none loses); not in the software-barrier builds (S2X 154/180 -4 %).
#ifndef DVBS2_PF_SMALL_MAX_DMAX
#define DVBS2_PF_SMALL_MAX_DMAX 8 // up to this degree class the whole record is double-buffered in scalar registers. Measured: class 8 loses 2-3 % without it
#endif                            // (B4 119.7 -> 117.1 k), class 12 GAINS 1-3 % without it (3/5, 2/3 normal, S2X 11/20, T2 2/3, short 2/3; short 3/5 -0.7 %)
#ifndef DVBS2_PREFETCH_AFTER_BARRIER_MAX_DMAX
#define DVBS2_PREFETCH_AFTER_BARRIER_MAX_DMAX 8
#endif
#ifndef DVBS2_PRETEST_CHUNK
#define DVBS2_PRETEST_CHUNK 1 // the one-layer syndrome pre-test four edges per trip (round 6; 0: edge by edge)
#endif
#ifndef DVBS2_WAIT_RECORDS
#define DVBS2_WAIT_RECORDS 1 // first measured on the plain class-8 build: B4 119.6 -> 115.9 k (off); re-measured once B4 ran the packed one-frame build: B4 133.2 -> 134.5 k, 2/5 normal +0.7 %, 3/5 +1 %, the others +-0.5 % (on)
#endif
#ifndef DVBS2_WAIT_BEFORE_STORE
#define DVBS2_WAIT_BEFORE_STORE 1
#endif
// s_waitcnt vmcnt(0) (expcnt, lgkmcnt untouched) that memory operations are not moved across
// (kWaitStore in scope: not in the builds with software frame barriers -- S2X 154/180 lost 4 % with it)
#define DVBS2_WAIT_VM0() do { if (kWaitStore) { asm volatile("" ::: "memory"); __builtin_amdgcn_s_waitcnt(0x0f70); asm volatile("" ::: "memory"); } } while (0)
#ifndef DVBS2_TLC_FWALK_MIN_DMAX
#define DVBS2_TLC_FWALK_MIN_DMAX 24 // the near pair of a two-level lane chain walked in float (six instructions per row, 16-byte operand records) in the
                                    // packed hazard nodes from this degree class up -- measured (round 5): 5/6 normal +3.1 %, 9/10 normal +0.35 %; 3/4 normal (class 16) -2.0 %
#endif

namespace dvbs2 {

// Thread mapping: a workgroup of 12 wavefronts decodes a PAIR of FECFRAMEs in lockstep; wavefronts 0-5 own
// frame 2b, wavefronts 6-11 own frame 2b+1. Inside a half, thread t (< 360) owns check row t of every
// circulant layer. Two frames are what the 160 KB of LDS hold (2 x (64800 + 9360) bytes for normal frames);
// putting them in ONE workgroup makes the hardware spread its 12 waves 3 per SIMD, all in the same phase of
// the same layer, so the per-layer barrier costs no load-imbalance wait (two independent 6-wave workgroups
// land 2,2,1,1 on the SIMDs and spend a quarter of their time waiting for the doubly loaded ones).
constexpr int kHalf = 384;          // threads per frame (6 wavefronts; threads 0..359 active)
constexpr int kThreads = 2 * kHalf; // 12 wavefronts
constexpr int kM = 360;
constexpr int kMsgStride = 384;     // message slots per (layer, word)
constexpr int kSvWords = 14;        // sign-vector dwords per 360-bit group (360 bits + 32-bit wrap extension, even for b64 stores)

// Layer record (uniform data, read with scalar loads): RS = 2*DMAX + 4 dwords.
//   word 0: cnt | sync_before << 15 | block << 16
//   words 4+2k, 5+2k (k < deg): entry k as  S0 = 360*g + rot  and  thr = 360 - rot
// Entry k addresses the LDS window [360*g, 360*g + 360) rotated by rot: check row j touches byte
// 360*g + (j + rot) mod 360 = (j < thr ? S0 + j : S0 + j - 360).
__host__ __device__ constexpr int rec_stride(int dmax) { return 2 * dmax + 4; }       // per-layer records (recs)
__host__ __device__ constexpr int rec_stride_wave(int dmax) { return 2 * dmax + 12; } // per-(layer, wave) records of the packed builds (wrecs)
// In FRONT of the per-layer records (recs[-kRecHeaderWords ..]): what the group-synchronous stop needs (group_decide) -- the base of the
// handle's `iters` array (the kernel's own `iters` argument minus it = the first frame of this launch), the base of the per-group words
// and the group size. Kept out of the kernel's argument list on purpose: arguments stay live in SGPRs for the whole kernel, and the
// one-frame builds of the degree class 16 answered three more of them with 25 more spilled scalars and 2-3 % (measured); here they are
// fetched with two scalar loads once per update, by the lane that reports.
constexpr int kRecHeaderWords = 8; // [0,1] iters base, [2,3] base of the per-frame status words (group_decide), [4] group size, [5] polls before a waiting member gives up, rest unused
// per frame: N LLR bytes, then the sign-vector area (syndrome test; scratch of the ordered hazard phases during a sweep:
// at least kChainScratchWords dwords, which is what short frames get instead of their small sign-vector area), then 8 flag words
#ifndef DVBS2_CHAIN_MAX_BLOCK
#define DVBS2_CHAIN_MAX_BLOCK 180 // round 4: 128 -> 180 (blocks 129..180 are three-step block-scheme layers otherwise): 3/4 normal +1.6 %, 3/5 +1.3 %, B4 / 2/5 normal +0.4 %
#endif
constexpr int kChainMaxBlock = DVBS2_CHAIN_MAX_BLOCK;                                             // largest block walked as a register chain
constexpr int kChainScratchWords = (kM + kChainMaxBlock) * 5 + 4;               // (360 + block) x (16-byte record + 4-byte log) + 16 bytes: the records are 16-byte aligned and the area starts at N, which is 8 mod 16 for short frames
__host__ __device__ constexpr int sv_area_words(int N) { return (N / kM) * kSvWords > kChainScratchWords ? (N / kM) * kSvWords : kChainScratchWords; }
__host__ __device__ constexpr size_t half_lds_bytes(int N) { return ((size_t)N + (size_t)sv_area_words(N) * 4 + 32 + 15) / 16 * 16; }

// EVERY access to LDS goes through a pointer whose TYPE says address space 3. A generic pointer that the compiler cannot trace back
// to the shared array (a function parameter, a pointer rebuilt from an integer for alignment, any `volatile` access) becomes a FLAT
// instruction: 64-bit address arithmetic, the long way round through the vector-memory path, and -- because flat loads return out of
// order with buffer loads -- an `s_waitcnt vmcnt(0) lgkmcnt(0)` at every use, which also waits for the message prefetch from HBM
// and serialises "rows in flight". Round 4 found the lane-chain walks, their operand tables and logs, the flag words and the
// software frame barrier all compiled that way (1 400 flat instructions in the degree class 8 alone): ~140 cycles per chain row.
typedef __attribute__((address_space(3))) uint8_t lds_byte_t;
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;
typedef __attribute__((address_space(3))) int lds_i32_t;
typedef __attribute__((address_space(3))) float lds_f32_t;
typedef float v4f32 __attribute__((ext_vector_type(4)));
typedef uint32_t v2u32 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) v4f32 lds_v4f_t;
typedef __attribute__((address_space(3))) v2u32 lds_v2u_t;
template <class T> __device__ __forceinline__ T* lds_align16(lds_u32_t* p) { return reinterpret_cast<T*>(((uint32_t)(size_t)p + 15u) & ~15u); }
__device__ __forceinline__ int wrap360(int t) { return t >= kM ? t - kM : t; }

// Barrier of ONE FRAME's six waves. The two frames of a workgroup share a CU only to get three waves on every SIMD
// (2+1 / 1+2: two separate 6-wave workgroups land 4,2,3,3 -- tools/ubench/placement.hip); nothing else couples them.
// With the hardware barrier both frames stall whenever either one is waiting for its slowest wave, for an LDS round trip
// or for an ordered hazard step; a barrier per frame lets the other frame's waves take the idle issue slots. gfx950 has
// no named barriers, so it is a counter in the frame's LDS region: every wave adds one (LDS executes a wave's operations
// in order, so its earlier writes are in place when the add lands) and polls until the count reaches the expected multiple of 6.
__device__ __forceinline__ void frame_barrier(volatile lds_i32_t* ctr, int& epoch, int lane)
{
    // (Round 4, with no FLAT access left in the kernel: the barrier without the wait for outstanding vector memory operations that
    // __syncthreads() implies -- s_waitcnt lgkmcnt(0) + s_barrier -- measured again: +-0.3 % on every BASELINE table. Not used.)
    if (!ctr) { __syncthreads(); return; } // hardware barrier of the workgroup (the default)
    epoch += 6;
    asm volatile("" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(const_cast<lds_i32_t*>(ctr), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    while (*ctr - epoch < 0) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}
#define lds_barrier() frame_barrier(hb_ctr, hb_epoch, hb_lane)
// The same where only LDS (the flag words) is handed over: without the wait for outstanding vector memory operations that
// __syncthreads() implies -- the group report of group_decide() is two fire-and-forget atomics whose acknowledgement would
// otherwise be waited for at the next barrier (short frames: 1.4 % of a sweep, measured).
__device__ __forceinline__ void frame_barrier_lds(volatile lds_i32_t* ctr, int& epoch, int lane)
{
    if (ctr) { frame_barrier(ctr, epoch, lane); return; }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
#define lds_only_barrier() frame_barrier_lds(hb_ctr, hb_epoch, hb_lane)

// Group-synchronous stopping rule. The reference decodes a SIMD batch of G frames in lockstep and stops the whole batch at the
// first update count at which EVERY lane passes the syndrome test (while (bad(any lane) && --trials >= 0),
// layered_decoder.hh:153). Frames of one group are dispatched together (consecutive workgroups) and run the same instruction
// stream, so they can simply agree after every test. Round 5: ONE STATUS WORD PER FRAME in global memory, written only by that frame,
//   status[f] = (update count of the test + 1) << 1 | passed          (0 = nothing reported yet; zeroed per call)
// A member reports with ONE relaxed store (fire and forget). A member that FAILS at count `it` knows the group goes on and does not
// wait. A member that PASSES reads the words of all members of its group (the first wave of the frame: lane m reads member m, one
// 256-byte access) and decides on them alone:
//   some member is PAST `it` (it only advances past a count at which a failure is known) or failed AT `it`   -> one more update (0)
//   every member passed AT `it`                                                                              -> stop (1)
//   else some member has not reported for `it` yet                                                           -> poll again
// Every member therefore leaves at the first count at which all pass -- the reference's count -- and no resume pass is needed.
// Each word has a single writer and its value only grows, so the decision needs no ordering BETWEEN words and no read-modify-write:
// rounds 3-4 kept {arrive, lastbad} per group, updated by a failing member with two separate relaxed atomics whose order at the L2
// the memory model does not promise; that dependence is gone. Relaxed at agent scope as before (a release /
// acquire at agent scope writes back and invalidates the per-XCD L2: 12 % of the never-converging batch, measured in round 3).
// Returns 2 = gave up waiting after spin_max polls (members not co-resident for milliseconds: never observed; the frame then stops
// at its own good point like in rounds 1-2 and the host-side resolution, ldpc_group_targets_kernel + resume launches, finishes the
// group -- a frame only ever advances past a count at which some member is known to have failed, so it can never overshoot).
// Called by ALL lanes of the frame's first wave (wave-uniform arguments); groups of at most 64 frames.
constexpr int kGroupSpinMax = 1 << 12; // polls of ~2 us
__device__ __forceinline__ int group_decide(int* st /*status words of this frame's group*/, int members, int me /*this frame's index in its group*/,
                                            int it, bool good, int lane, int spin_max = kGroupSpinMax)
{
    const int mine = ((it + 1) << 1) | (good ? 1 : 0);
    // (measured against an atomic max without a returned value -- the same thing for a value that only grows --: identical rates on
    // short 1/4, 2/5, 1/4 normal, B4 and medium 1/5, where a frame reports every 40-90 us: notes/r05_experiments.md)
    if (lane == 0) __hip_atomic_store(st + me, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!good) return 0; // (nothing is waited for)
    for (int spin = 0;; spin++) {
        int s = mine; // lanes without a member, and this frame itself (its own store need not be visible to its own load yet), are neutral
        if (lane < members && lane != me) s = __hip_atomic_load(st + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const bool goes_on = s > mine || s == mine - 1; // a later count, or this count failed
        const bool missing = s < mine - 1;              // an earlier count (or nothing yet)
        if (__ballot(goes_on) != 0) return 0;
        if (__ballot(missing) == 0) return 1;
        if (spin >= spin_max) return 2;
        __builtin_amdgcn_s_sleep(8);
    }
}

// Pinned instruction selection for the two spots where the compiler's canonical form costs more issue slots.
// clamp(a + b + 128, 0, 255) as v_add3_u32 + v_med3_i32 (the compiler emits add, max, add, min).
__device__ __forceinline__ int sat_sum_u8(int a, int b)
{
    int r; // one asm statement: between two the compiler puts an s_nop (it cannot see that the pair has no hazard)
    asm("v_add3_u32 %0, %1, %2, %3\n\tv_med3_i32 %0, %0, 0, %4" : "=&v"(r) : "v"(a), "v"(b), "s"(128), "s"(255));
    return r;
}
// R2: mag = clamp(|Lb - mb| - 1, 0, 126) as v_sad_u16 + v_med3_i32 (the compiler splits the clamp into max + min)
__device__ __forceinline__ int mag_offset(int Lb, int mb)
{
    int r;
    const int a = (int)__builtin_amdgcn_sad_u16((uint32_t)Lb, (uint32_t)mb, 0xffffffffu);
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(a), "s"(126));
    return r;
}
// LLR bytes are addressed with ABSOLUTE LDS addresses (the row index carries the frame's offset and the address of
// the dynamic LDS array): going through the array symbol costs one `v_add_u32 v, <lds_all>, v` per access, because the
// array's address is a link-time constant the compiler cannot fold into an address that inline asm produced.
__device__ __forceinline__ int lds_rd(int a) { return *reinterpret_cast<const lds_byte_t*>((size_t)(uint32_t)a); }
__device__ __forceinline__ void lds_wr(int a, int v) { *reinterpret_cast<lds_byte_t*>((size_t)(uint32_t)a) = (uint8_t)v; }
// Round 5: the builds with packed nodes keep the LLR bytes in LDS as TWO'S COMPLEMENT (TC): the packed arithmetic works on value << 8 in signed
// 16-bit halves, and with offset-binary bytes every pair paid one xor after its reads and one before its writes (8 of the ~110 VALU instructions
// of a degree-7 check). The scalar paths of such a build (layer 0, the ordered phase of hazard layers, waves whose record does not fit the
// packed format) convert at their LDS accesses; messages, the state in HBM and every other build stay as they were.
// Measured (interleaved A/B, all packed builds with and without): the degree class 8 gains (B4 135.4 -> 137.7 k, S2X 9/20 +0.7 %), every other
// class loses 0.2-3 % (3/4 normal -2.7 %, 4/5 -3 %, short 3/4 -2.9 %: their scalar paths pay the conversion, and removing 7 % of the packed
// node's VALU instructions buys almost nothing where the layer is as much bound by its message traffic and barriers) -- so: class 8 only.
#ifndef DVBS2_TC_MAX_DMAX
#define DVBS2_TC_MAX_DMAX 8
#endif
template <bool TC> __device__ __forceinline__ int lds_rdx(int a) { const int v = lds_rd(a); return TC ? (v ^ 0x80) : v; }          // offset-binary value of the LLR byte at a
template <bool TC> __device__ __forceinline__ void lds_wrx(int a, int v) { lds_wr(a, TC ? (v ^ 0x80) : v); }                        // store an offset-binary value
template <bool TC> constexpr uint32_t kObPair = TC ? 0u : 0x80008000u; // offset binary -> two's complement << 8 of a pair register
__device__ __forceinline__ int lds_address_of(const uint8_t* p) { return (int)(uint32_t)(size_t)(const lds_byte_t*)p; }

// LDS address of check row jj for entry (S0 = 360*g + rot, thr = 360 - rot): S0 + jj, minus 360 when jj >= thr.
// The canonical compare + select + add3 is three half-rate VALU instructions; this is four full-rate ones (2.5 vs 4.3
// cycles each on gfx950): subtract, sign mask, bitfield select between jj and jj - 360 (v_bitop3), add.
__device__ __forceinline__ int wrap_addr(int jj, int jjb, int jjb360, uint32_t S0, uint32_t thr)
{
    // jjb = jj + byte offset of this frame's LDS region (folded in here: no separate base add per access)
    int r;
    asm("v_subrev_u32 %0, %4, %1\n\t"
        "v_ashrrev_i32 %0, 31, %0\n\t"
        "v_bitop3_b32 %0, %0, %2, %3 bitop3:0xca\n\t"
        "v_add_u32 %0, %5, %0"
        : "=&v"(r) : "v"(jj), "v"(jjb), "v"(jjb360), "s"(thr), "s"(S0));
    return r;
}

// R3: the two smallest of N magnitudes. The kernel is bound by the VALU pipe and min/max/med3 are half-rate
// there, so the count matters: triples go through v_min3 + v_med3 (smallest and second smallest of three in two
// instructions), two sorted pairs merge in three, a single value folds in with two -- 9 instructions for seven
// values where the running (min0, min1) update needs 14.
__device__ __forceinline__ int vmin3_i32(int a, int b, int c) { int r; asm("v_min3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
__device__ __forceinline__ int vmed3_i32(int a, int b, int c) { int r; asm("v_med3_i32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }
template <int N>
__device__ __forceinline__ void two_smallest(const int* v, int& m0, int& m1)
{
    static_assert(N >= 1, "empty set");
    int k;
    if constexpr (N == 1) { m0 = v[0]; m1 = 127; k = 1; }
    else if constexpr (N == 2) { m0 = min(v[0], v[1]); m1 = max(v[0], v[1]); k = 2; }
    else { m0 = vmin3_i32(v[0], v[1], v[2]); m1 = vmed3_i32(v[0], v[1], v[2]); k = 3; }
#pragma unroll
    for (; k + 3 <= N; k += 3) {
        const int g0 = vmin3_i32(v[k], v[k + 1], v[k + 2]), g1 = vmed3_i32(v[k], v[k + 1], v[k + 2]);
        const int t = max(m0, g0);
        m0 = min(m0, g0);
        m1 = vmin3_i32(t, m1, g1);
    }
#pragma unroll
    for (; k < N; k++) { m1 = vmed3_i32(m0, m1, v[k]); m0 = min(m0, v[k]); }
}

// R2 without its clamp: |Lb - mb| - 1 in [-1, 254]. Clamping to [0, 126] is monotone, so the two smallest clamped
// magnitudes are the clamps of the two smallest raw ones (two v_med3 per check instead of one per edge), and the
// selection "mag == min0 ? min1 : min0" becomes min0 + min1 - med3(raw, min0, min1): clamping raw into
// [min0, min1] gives min0 exactly when the clamped magnitude is the smallest one.
__device__ __forceinline__ int mag_raw(int Lb, int mb) { return (int)__builtin_amdgcn_sad_u16((uint32_t)Lb, (uint32_t)mb, 0xffffffffu); }
__device__ __forceinline__ int clamp_mag(int x) { int r; asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(x), "s"(126)); return r; }
constexpr int kMagAbsent = 0x7fff; // a link that does not exist (check (0,0)): above every raw magnitude

// low bytes of four 32-bit values -> one dword (two v_perm_b32 + or)
__device__ __forceinline__ uint32_t pack4_lo8(int a, int b, int c, int d)
{
    const uint32_t lo = __builtin_amdgcn_perm((uint32_t)b, (uint32_t)a, 0x0c0c0400u); // a.b0 | b.b0 << 8
    const uint32_t hi = __builtin_amdgcn_perm((uint32_t)d, (uint32_t)c, 0x04000c0cu); // c.b0 << 16 | d.b0 << 24
    return lo | hi;
}

// One check node (layered_decoder.hh:56-77 + algorithms.hh:170-192,203-206), fully unrolled for its degree.
// LLRs are offset-binary bytes Lb = L + 128 in LDS; messages are offset-binary bytes, 4 per dword.
// The kernel is VALU-issue bound (not HBM bound): ~22 VALU + 2 LDS instructions per edge.
//
// Parity links. Classic layout (PR = false): both parity LLRs live in LDS like the data LLRs. "Parity in records"
// (PR = true, low-rate tables, see ldpc_kernel_pr.hpp): parity row i is only ever touched by thread j of layers i and
// i+1, so it never needs LDS -- the own-parity LLR arrives in `own_in` (byte 7 of the NEXT layer's message
// record, where layer i+1 left it in the previous sweep), the previous-parity LLR is `carry` (what this thread's
// own-parity link produced one layer ago), the new own-parity LLR becomes the carry and the new previous-parity
// LLR is returned in byte 7 of this layer's record. Only row q-1 (own parity of the LAST layer, previous
// parity of layer 0 shifted by one lane) stays in LDS.
template <int DEG, bool LAYER0, bool PR = false, bool LAST = false, bool TC = false>
__device__ __forceinline__ void check_node(uint8_t* __restrict__ lds /*the whole LDS array*/, const uint32_t* ent /*uniform: S0, thr pairs*/,
                                           int jj, int lb /*byte offset of this frame's region*/, const uint32_t* mw, uint32_t* nm,
                                           int own_in = 0, int* carry = nullptr)
{
    constexpr bool OWN_REG = PR && !LAST;     // entry DEG-2
    constexpr bool PREV_REG = PR && !LAYER0;  // entry DEG-1
    // Issue priority RISES as the wave advances through the node (0 while it computes addresses and issues its LDS
    // reads, 1 for the reduction, 3 from the output phase until the next node starts): a wave that holds its data
    // is served before one that is about to wait for LDS anyway. Measured on B4: classic kernel 95.5 k -> 103.5 k
    // frames/s, parity-in-records 104.4 k -> 105.5 k; the opposite order costs 8 %.
    __builtin_amdgcn_s_setprio(0);
    int ad[DEG], Lb[DEG];
    const int jjb = jj + lb, jjb360 = jjb - kM;
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        // address = S0 + jj, minus 360 when jj >= thr; the two parity entries have rot = 0 (never wrap) except
        // the previous-parity entry of layer 0 (rot = 359)
        if (k >= DEG - 2 && !(LAYER0 && k == DEG - 1)) ad[k] = jjb + (int)ent[2 * k];
        else ad[k] = wrap_addr(jj, jjb, jjb360, ent[2 * k], ent[2 * k + 1]);
    }
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        if (OWN_REG && k == DEG - 2) Lb[k] = own_in;
        else if (PREV_REG && k == DEG - 1) Lb[k] = *carry;
        else Lb[k] = lds_rdx<TC>(ad[k]);
    }
    // check (0,0) has no previous-parity link (layered_decoder.hh:56,63-66)
    const bool last_valid = !LAYER0 || jj != 0;
    int spare = 0x80;

    int inp[DEG], mg[DEG];
    int min0 = 127, min1 = 127, signs = 0;
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        const int mb = (int)((mw[k >> 2] >> (8 * (k & 3))) & 0xffu);
        // R1 inp = sat8(L - m); R2 mag = usat(qabs(inp) - 1) == med3(|L - m| - 1, 0, 126)
        int d = min(max(Lb[k] - mb, -128), 127);
        int mag = mag_raw(Lb[k], mb);
        if (LAYER0 && k == DEG - 1) { d = last_valid ? d : 0; mag = last_valid ? mag : kMagAbsent; }
        inp[k] = d; mg[k] = mag;
        signs ^= d; // R4 xor of the sign bits
    }
    __builtin_amdgcn_s_setprio(1);
    two_smallest<DEG>(mg, min0, min1); // R3 on raw magnitudes; R2's clamp once per check
    min0 = clamp_mag(min0); min1 = clamp_mag(min1);
    const int s01 = min0 + min1;
    int msgc[4 * ((DEG + 3) / 4)];
#pragma unroll
    for (int k = 0; k < 4 * ((DEG + 3) / 4); k++) msgc[k] = 0;
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        // R5 out = vsign(mag == min0 ? min1 : min0, (signs ^ x) | 127); mag is min0 or >= min1, so the selected
        // magnitude is min0 + min1 - min(mag, min1)
        const int other = s01 - vmed3_i32(mg[k], min0, min1);
        const int sg = (signs ^ inp[k]) >> 31;
        const int out = (other ^ sg) - sg;
        // R6 LLR = sat8(inp + out) with the unclamped out; R7 stored message = clamp(out, -32, 31)
        const int nl = sat_sum_u8(inp[k], out);
        if (OWN_REG && k == DEG - 2) *carry = nl;
        else if (PREV_REG && k == DEG - 1) spare = nl;
        else if (!(LAYER0 && k == DEG - 1) || last_valid) lds_wrx<TC>(ad[k], nl);
        msgc[k] = min(max(out, -32), 31);
    }
    __builtin_amdgcn_s_setprio(3);
    // two's-complement low bytes ^ 0x80 = offset binary
#pragma unroll
    for (int w = 0; w < (DEG + 3) / 4; w++)
        nm[w] = pack4_lo8(msgc[4 * w], msgc[4 * w + 1], msgc[4 * w + 2], msgc[4 * w + 3]) ^ 0x80808080u;
    if (PR) { // byte 7 of the record carries the previous-parity LLR (DEG <= 7)
        const uint32_t w1 = ((DEG + 3) / 4 > 1) ? nm[1] : 0x80808080u;
        nm[1] = (w1 & 0x00ffffffu) | ((uint32_t)spare << 24);
    }
}

// two-level lane chain (check_node_hazard): the degree class 32 without the heavy-hazard paths (9/10 normal); not in the builds with software
// frame barriers, which only tables without hazard layers run (S2X 154/180 lost 2.5 % to the larger kernel)
// Which degree classes carry it is MEASURED (MI355X, interleaved A/B of whole tables, notes/r03_experiments.md): at run time the chain is
// never slower than the ordered steps it replaces (9/10 normal + 13 %, 3/5 normal + 10 %, short 5/6 + 4 %, 3/4 normal + 2.4 %), but
// compiling it in costs the packed / one-frame builds of the classes 12 and 28 eight percent on every table (2/3, T2 2/3, 8/9 normal)
// and the class 20 what its one table gains -- so: 16 (3/4 normal + 5 % net, short 5/6 + 2 %, short 2/3 - 4 %), 24 (5/6 normal + 1.3 %),
// 32 (9/10 normal + 8 %).
template <int DMAX, bool HZ2> constexpr bool kTlc = (DMAX == 16 || DMAX == 24 || DMAX == 32) && !HZ2;
// Hazard layers with the packed first / last phase (check_node_hazard<..., V2P>, round 5): compiled into the packed builds of the degree
// classes from DVBS2_V2P_MIN_DMAX up -- the classes whose hazard layers all took the plain node (no packed chain node there).
#ifndef DVBS2_V2P_MIN_DMAX
#define DVBS2_V2P_MIN_DMAX 20
#endif
__host__ __device__ constexpr bool v2p_class(int dmax) { return dmax >= DVBS2_V2P_MIN_DMAX; }
#ifndef DVBS2_V2_PURE_MIN_DMAX
#define DVBS2_V2_PURE_MIN_DMAX 32 // measured (round 5, interleaved A/B): class 32 (9/10 normal) 80.2 -> 83.9 k; 28 (8/9) 95.9 -> 91.2 k, 24 (5/6) 76.9 -> 74.2 k
#endif
__host__ __device__ constexpr bool v2_pure_class(int dmax) { return dmax >= DVBS2_V2_PURE_MIN_DMAX; }
__host__ __device__ constexpr bool tlc_class(int dmax) { return dmax == 16 || dmax == 24 || dmax == 32; }
constexpr int kTlcLowRegMinDmax = 24; // from this degree class on a two-level-chain layer keeps its regular entries in the low-register form
__device__ __forceinline__ int pm_pack(int magp, int d) { return (int)__builtin_amdgcn_perm((uint32_t)magp, (uint32_t)d, 0x0c0c0400u); } // d.b0 | magp.b0 << 8
__device__ __forceinline__ int pm_inp(int pm) { return __builtin_amdgcn_sbfe(pm, 0, 8); }
__device__ __forceinline__ int pm_min_clamped(int p) { return clamp_mag((int)((uint32_t)p >> 8) - 1); } // R2 on a minimum: clamp(|x| - 1, 0, 126)
// ---------------------------------------------------------------------------------------------------------------------------------
// Packed check node ("v2", regular layers other than layer 0). The sweep is bound by VALU issue slots, so the node is
// built to need fewer of them per edge:
//   * ADDRESSES. Records are per WAVE (the host knows which 64 rows a wave owns): for an entry whose wrap point lies
//     outside the wave's rows the window offset is pre-adjusted (S0 or S0 - 360) and the address is ONE add; the few
//     entries whose wrap point falls inside the wave ("mixed", on average deg / 6) sit in the first NFIX slots and get
//     + 360 on the lanes below the wrap point under an EXEC mask taken from the record (one more add). 1.3 instead of 4
//     VALU instructions per edge.
//   * ARITHMETIC on PAIRS of edges in the halves of one register, as value << 8 in signed 16 bit: the saturating packed
//     add / subtract then IS the reference's int8 saturation (R1 sat8(L - m), R6 sat8(inp + out)), the message clamp (R7)
//     is one packed max + min per pair, |inp| is packed max(d, 0 - d). A positive saturation leaves 0xff in the low
//     byte of a half; nothing below lets it reach a result (see the notes at the uses).
//   * MAGNITUDES are reduced in scalar form (v_min3 / v_med3 triples need fewer slots than a packed running pair), the
//     selection "mag == min0 ? min1 : min0" is packed again: T - clamp(mag, B0, B1) with B0 = min0, B1 = B0 + (min1' -
//     min0'), T = min1' + B0, where x' = max(x - 1, 0) (R2's offset and floor applied once per check, not per edge).
// Messages of such a layer are two's complement bytes (this layer's records are private to it: layer 0 and hazard
// layers keep offset binary); logical entry e = 2 j + h of pair j lives in dword j / 2, byte (j & 1) + 2 h, so that both
// pairs of a dword unpack with one instruction each. LLR bytes in LDS stay offset binary (shared with the other paths).
typedef short v2s16 __attribute__((ext_vector_type(2)));
typedef unsigned short v2u16 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2s16 as_v2s(uint32_t x) { return __builtin_bit_cast(v2s16, x); }
__device__ __forceinline__ uint32_t as_u32(v2s16 x) { return __builtin_bit_cast(uint32_t, x); }
__device__ __forceinline__ void lds_wr_hi(int a, uint32_t v) { *reinterpret_cast<lds_byte_t*>((size_t)(uint32_t)a) = (uint8_t)(v >> 16); } // ds_write_b8_d16_hi

// + 360 on the lanes of `mask` (EXEC is saved and restored: the statement is valid under any execution mask)
__device__ __forceinline__ int fix_wrap(int ad, uint32_t mlo, uint32_t mhi)
{
    // (readfirstlane: a no-op for a value that already sits in an SGPR, and it keeps an SGPR that the register allocator
    // spilled to a VGPR lane from being handed to the scalar instruction as a VGPR)
    const unsigned long long mask = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((int)mhi) << 32) | (uint32_t)__builtin_amdgcn_readfirstlane((int)mlo);
    unsigned long long save;
    asm volatile("s_mov_b64 %1, exec\n\ts_and_b64 exec, exec, %2\n\tv_add_u32 %0, 0x168, %0\n\ts_mov_b64 exec, %1"
                 : "+v"(ad), "=&s"(save) : "s"(mask) : "scc");
    return ad;
}

#ifndef DVBS2_V2_NFIX32
#define DVBS2_V2_NFIX32 10 // fix slots of the degree class 32: with 10 every wave record of S2X 154/180 fits the packed format (8: two of its 150 did not)
#endif
__host__ __device__ constexpr int v2_nfix(int dmax) { return dmax == 32 ? DVBS2_V2_NFIX32 : dmax / 4; } // fix slots per record: masks live in record words 4 + dmax + 2 k

// Message storage of the packed nodes. A stored message is clamp(out, -32, 31) (R7): six bits. P6 = true keeps them as six-bit
// two's complement fields, five per dword (entry e in word e / 5 at bit 6 (e % 5); a last word with one or two fields is a
// 16-bit access): 4, 4, 6, 6, 8 ... bytes per check for degree 4, 5, 6, 7, 8 instead of 8. The regular layers of the
// low-degree tables run at the bandwidth the memory system gives to this access pattern (~4.9 TB/s of message traffic on
// table B4, whatever the node costs), so the bytes are what counts there; the unpacking costs ~2.5 VALU instructions per
// edge, which those layers have to spare. P6 = false: one byte per message (pair j in bytes (j & 1) and (j & 1) + 2 of word j / 2).
template <bool P6>
__device__ __forceinline__ uint32_t msg_pair16(const uint32_t* mw, int j)
{
    if constexpr (P6) {
        const int e0 = 2 * j, e1 = 2 * j + 1;
        const int lo = __builtin_amdgcn_sbfe((int)mw[e0 / 5], 6 * (e0 % 5), 6), hi = __builtin_amdgcn_sbfe((int)mw[e1 / 5], 6 * (e1 % 5), 6);
        return __builtin_amdgcn_perm((uint32_t)hi, (uint32_t)lo, 0x040c000cu); // [hi << 8 | lo << 8]; a field past the degree reads 0
    } else {
        const uint32_t w = mw[j >> 1];
        return (j & 1) ? (w & 0xff00ff00u) : __builtin_amdgcn_perm(w, 0u, 0x060c040cu); // bytes 2, 0 of w to bytes 3, 1
    }
}
template <bool P6, int NP, int NW>
__device__ __forceinline__ void msg_pack16(const uint32_t* R /*clamped messages << 8 in both halves, pad half zero*/, uint32_t* nm)
{
    if constexpr (P6) {
#pragma unroll
        for (int w = 0; w < NW; w++) nm[w] = 0;
#pragma unroll
        for (int e = 0; e < 2 * NP; e++) {
            if (e / 5 < NW) {
                const uint32_t f = __builtin_amdgcn_ubfe(R[e >> 1], (e & 1) ? 24 : 8, 6);
                nm[e / 5] = (e % 5) ? ((f << (6 * (e % 5))) | nm[e / 5]) : f;
            }
        }
    } else {
#pragma unroll
        for (int w = 0; w < (NP + 1) / 2; w++)
            nm[w] = (2 * w + 1 < NP) ? ((R[2 * w] >> 8) | R[2 * w + 1]) : (R[2 * w] >> 8);
    }
}
// words of one check's message record that hold fields, and whether word k is a 16-bit access, for degree deg
__host__ __device__ constexpr int p6_fields(int deg, int k) { return deg - 5 * k < 0 ? 0 : (deg - 5 * k > 5 ? 5 : deg - 5 * k); }

template <int DEG, int DMAX, bool P6, bool TC, class Prefetch>
__device__ __forceinline__ void check_node_v2(const uint32_t* ent /*record words 4..: S0w[DMAX], then (mask lo, mask hi)[NFIX]*/,
                                              int jjb, const uint32_t* mw, uint32_t* nm, Prefetch prefetch_next_record)
{
    constexpr int NP = (DEG + 1) / 2;      // pairs
    constexpr int NFIX = v2_nfix(DMAX) < DEG - 2 ? v2_nfix(DMAX) : DEG - 2; // parity entries never wrap
    constexpr bool ODD = (DEG & 1) != 0;   // the upper half of the last pair is a pad: L = m = 0, magnitude "absent"
    __builtin_amdgcn_s_setprio(0);
    int ad[DEG];
    auto addresses = [&]() {
#pragma unroll
        for (int k = 0; k < DEG; k++) ad[k] = jjb + (int)ent[k];
#pragma unroll
        for (int k = 0; k < NFIX; k++) ad[k] = fix_wrap(ad[k], ent[DMAX + 2 * k], ent[DMAX + 2 * k + 1]);
    };
    addresses();
#ifndef DVBS2_V2_KEEP_AD_MAXDEG
#define DVBS2_V2_KEEP_AD_MAXDEG 32 // experiments: above this degree the addresses are computed again in the output phase instead of held across the node
#endif
    constexpr bool KEEP_AD = DEG <= DVBS2_V2_KEEP_AD_MAXDEG;
    int Lb[DEG];
#pragma unroll
    for (int k = 0; k < DEG; k++) Lb[k] = lds_rd(ad[k]);
    v2s16 d[NP], a[NP];
    uint32_t sx = 0;
#pragma unroll
    for (int j = 0; j < NP; j++) {
        const uint32_t M = msg_pair16<P6>(mw, j); // messages of pair j: << 8 in both halves
        const uint32_t hi = (ODD && j == NP - 1) ? (TC ? 0x00u : 0x80u) : (uint32_t)Lb[2 * j + 1];
        const uint32_t L = __builtin_amdgcn_perm(hi, (uint32_t)Lb[2 * j], 0x040c000cu) ^ kObPair<TC>; // -> two's complement << 8
        d[j] = __builtin_elementwise_sub_sat(as_v2s(L), as_v2s(M));           // R1 (a half that saturates upwards reads 0x7fff)
        sx ^= as_u32(d[j]);                                                   // R4: bits 15 and 31 collect the signs
        a[j] = __builtin_elementwise_max(d[j], __builtin_elementwise_sub_sat(as_v2s(0u), d[j])); // |inp| << 8 (0x7fff for -128 and for saturated halves)
    }
    __builtin_amdgcn_s_setprio(1);
    // (Issuing the NEXT layer's scalar record loads from this point -- scalar memory shares its counter with LDS, so a load in
    // flight turns every LDS wait into "wait for everything" -- was tried with a scheduling barrier and an ordering dependency:
    // it cost 9 % on table B4 and a factor 4 on the degree-30 class through what it does to register allocation. The loads stay
    // at the top of the layer; notes/history.md 3.4.)
    (void)prefetch_next_record;
    int mg[DEG];
#pragma unroll
    for (int k = 0; k < DEG; k++) mg[k] = (k & 1) ? (int)(as_u32(a[k >> 1]) >> 16) : (int)(as_u32(a[k >> 1]) & 0xffffu);
    int n0, n1;
    two_smallest<DEG>(mg, n0, n1);
    // the low byte (0xff after a saturation) is dropped HERE, once per check: every selected magnitude below is B0/B1-clamped,
    // so a 0x7fff among the inputs can only come out as the clean 0x7f00 level it stands for
    n0 &= 0x7f00; n1 &= 0x7f00;
    // R2: mag = max(|inp| - 1, 0), applied to the two minima (monotone); unsigned saturating subtract (full rate)
    const int n0m = (int)__builtin_elementwise_sub_sat((uint32_t)n0, 256u), n1m = (int)__builtin_elementwise_sub_sat((uint32_t)n1, 256u);
    const int B0 = n0, B1 = n0 + n1m - n0m, T = n1m + n0;
    const v2s16 B0p = { (short)B0, (short)B0 }, B1p = { (short)B1, (short)B1 }, Tp = { (short)T, (short)T };
    const uint32_t tm = (uint32_t)((int)(sx ^ (sx << 16)) >> 31); // all ones when the number of negative inputs is odd
    if constexpr (!KEEP_AD) { asm volatile("" ::: "memory"); addresses(); }
    uint32_t R[NP];
#pragma unroll
    for (int j = 0; j < NP; j++) {
        // R5: |out| = mag == min0 ? min1' : min0'  ==  T - clamp(|inp|, B0, B1); sign = total ^ own
        const v2s16 c = __builtin_elementwise_min(__builtin_elementwise_max(a[j], B0p), B1p);
        const v2s16 other = Tp - c;
        const v2s16 sg = as_v2s(as_u32(d[j]) ^ tm) >> (v2s16){ 15, 15 };
        const v2s16 out = as_v2s(as_u32(other) ^ as_u32(sg)) - sg;
        // R6: LLR = sat8(inp + out); the low byte of a half never reaches the byte that is stored
        const uint32_t nl = (as_u32(__builtin_elementwise_add_sat(d[j], out)) ^ kObPair<TC>) >> 8;
        lds_wr(ad[2 * j], (int)nl);
        if (!(ODD && j == NP - 1)) lds_wr_hi(ad[2 * j + 1], nl);
        // R7
        R[j] = as_u32(__builtin_elementwise_min(__builtin_elementwise_max(out, (v2s16){ -32 * 256, -32 * 256 }), (v2s16){ 31 * 256, 31 * 256 }));
    }
    __builtin_amdgcn_s_setprio(3);
    if (ODD) R[NP - 1] &= 0x0000ffffu; // the pad's message stays zero
    msg_pack16<P6, NP, DMAX / 4>(R, nm);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Hazard layer with ONE pair (two entries X, Y of one group, block B <= kChainMaxBlock), packed arithmetic, register chain.
// The host orders the pair so that X's bit of row r is Y's bit of row r + B: row r hands its new X value to row r + B.
//   heads  r < B            X and Y both original                       -> Y written at once, X starts the chain
//   middle B <= r < 360-B   X original, Y = X of row r - B (chain)
//   tails  r >= 360 - B     X = Y of head r + B - 360 (in LDS after the heads), Y from the chain; final writer of X
// Phases (frame barriers between them): P1 all rows: regular entries read and reduced (packed, as check_node_v2); heads also
// resolve their pair. P2 rows >= B: read X, publish the chain operands of their row as four floats. P3 the B head lanes walk
// r -> r + B: six VALU instructions per step on exact small integers in float (fma, two med3 with a negated operand, sub,
// add, clamp) plus one 16-byte read and one 4-byte log write; a lone wave issues one instruction per ~4 cycles, so the step
// costs its instruction count. P4 all rows: pair completed from the log, final minima, outputs of every entry, messages.
// Identical to the reference's row order: a bit of the pair is touched by exactly two rows, the later one sees the earlier one.
//   chain step for row r with incoming Y value c (offset binary 0..255):  x = sigma (c - 128 - mY),
//   out = sgn(x) min(P, max(|x| - 1, 0)) = w - sgn(w), w = clamp(x, -(P+1), P+1);  c' = clamp(inpX + 128 + out, 0, 255)
__device__ __forceinline__ float as_f32(uint32_t x) { return __builtin_bit_cast(float, x); }
__device__ __forceinline__ float vmed3_f32(float a, float b, float c) { return __builtin_amdgcn_fmed3f(a, b, c); }
__device__ __forceinline__ float byte1_f32(uint32_t x) { return (float)((x >> 8) & 0xffu); } // v_cvt_f32_ubyte1

template <int DEG, int DMAX, bool P6, bool TC>
__device__ __forceinline__ void check_node_chain_v2(const uint32_t* ent /*S0w[DMAX], masks[NFIX + 2]*/, int jj, int jjb, bool work, int B,
                                                    const uint32_t* mw, uint32_t* nm, lds_u32_t* tab /*LDS scratch, 16-byte aligned*/,
                                                    volatile lds_i32_t* hb_ctr, int& hb_epoch, const int hb_lane)
{
    constexpr int NP = (DEG + 1) / 2;
    constexpr int NFIXH = (v2_nfix(DMAX) + 2) < DEG - 2 ? (v2_nfix(DMAX) + 2) : DEG - 2;
    constexpr bool ODD = (DEG & 1) != 0;
    constexpr bool KEEP_AD = DEG <= 16; // high degrees recompute the addresses in P4 instead of holding 30 registers across the phases
    lds_v4f_t* rec = reinterpret_cast<lds_v4f_t*>(tab);           // [360 + B] chain operands
    lds_f32_t* logv = reinterpret_cast<lds_f32_t*>(tab) + 4 * (kM + kChainMaxBlock); // [360 + B] value that arrived at row r
    const bool head = work && jj < B, body = work && jj >= B;
    const bool middle = body && jj + B < kM; // rows whose new X value travels down the chain; the others (tails) end a chain
    int LbX = 0x80;
    int ad[DEG];
    v2s16 d[NP], a[NP];
    uint32_t sxp = 0;
    int p0 = 0x7fff, p1 = 0x7fff;
    int Pm = 0;
    float c = 0.f;
    __builtin_amdgcn_s_setprio(0);
    auto addresses = [&]() {
#pragma unroll
        for (int k = 0; k < DEG; k++) ad[k] = jjb + (int)ent[k];
#pragma unroll
        for (int k = 0; k < NFIXH; k++) ad[k] = fix_wrap(ad[k], ent[DMAX + 2 * k], ent[DMAX + 2 * k + 1]);
    };
    auto pair0 = [&](int LbX, int LbY) { // d, |d| of the pair [X | Y] (LLR bytes as they lie in LDS)
        const uint32_t L = __builtin_amdgcn_perm((uint32_t)LbY, (uint32_t)LbX, 0x040c000cu) ^ kObPair<TC>;
        const uint32_t M = msg_pair16<P6>(mw, 0);
        d[0] = __builtin_elementwise_sub_sat(as_v2s(L), as_v2s(M));
        a[0] = __builtin_elementwise_max(d[0], __builtin_elementwise_sub_sat(as_v2s(0u), d[0]));
    };
    if (work) {
        addresses();
        int Lb[DEG];
#pragma unroll
        for (int k = 2; k < DEG; k++) Lb[k] = lds_rd(ad[k]);
        int LbY = 0x80;
        if (head) { LbX = lds_rd(ad[0]); LbY = lds_rd(ad[1]); }
        else if (middle) LbX = lds_rd(ad[0]); // original value: the only other row that touches this bit comes later (row jj + B)
#pragma unroll
        for (int j = 1; j < NP; j++) {
            const uint32_t M = msg_pair16<P6>(mw, j);
            const uint32_t hi = (ODD && j == NP - 1) ? (TC ? 0x00u : 0x80u) : (uint32_t)Lb[2 * j + 1];
            const uint32_t L = __builtin_amdgcn_perm(hi, (uint32_t)Lb[2 * j], 0x040c000cu) ^ kObPair<TC>;
            d[j] = __builtin_elementwise_sub_sat(as_v2s(L), as_v2s(M));
            sxp ^= as_u32(d[j]);
            a[j] = __builtin_elementwise_max(d[j], __builtin_elementwise_sub_sat(as_v2s(0u), d[j]));
        }
        int mg[DEG - 2];
#pragma unroll
        for (int k = 2; k < DEG; k++) mg[k - 2] = (k & 1) ? (int)(as_u32(a[k >> 1]) >> 16) : (int)(as_u32(a[k >> 1]) & 0xffffu);
        two_smallest<DEG - 2>(mg, p0, p1); // raw |inp| << 8 of the regular entries (a pad never enters: DEG - 2 real values)
        Pm = (int)(__builtin_elementwise_sub_sat((uint32_t)(p0 & 0x7f00), 256u) >> 8); // R2 on the partial minimum: 0..126
        if (head) {
            pair0(LbX, LbY);
            // the other entry of the pair is the only input outside the partial: |out_X| = min(Pm, mag Y) and vice versa
            const uint32_t sw = __builtin_amdgcn_alignbit(as_u32(a[0]), as_u32(a[0]), 16) & 0x7f007f00u;
            const v2u16 mgs = __builtin_elementwise_sub_sat(__builtin_bit_cast(v2u16, sw), (v2u16){ 256, 256 });
            const v2s16 other = __builtin_elementwise_min(__builtin_bit_cast(v2s16, mgs), (v2s16){ (short)(Pm << 8), (short)(Pm << 8) });
            const uint32_t par = (uint32_t)((int)(sxp ^ (sxp << 16)) >> 31);
            const uint32_t ds = __builtin_amdgcn_alignbit(as_u32(d[0]), as_u32(d[0]), 16);
            const v2s16 sg = as_v2s(ds ^ par) >> (v2s16){ 15, 15 };
            const v2s16 out = as_v2s(as_u32(other) ^ as_u32(sg)) - sg;
            const uint32_t raw = as_u32(__builtin_elementwise_add_sat(d[0], out));
            const uint32_t nl = raw ^ 0x80008000u;                      // offset binary in bytes 1 and 3 (the chain walks offset-binary values)
            lds_wr_hi(ad[1], (TC ? raw : nl) >> 8);                    // Y now (a tail row reads it as its X)
            c = byte1_f32(nl);                   // X starts the chain
        }
    }
    // chain operands of the middle rows (a tail row only receives: the walker logs what arrives there and needs nothing from it)
    if (middle) {
        const uint32_t L = __builtin_amdgcn_perm(TC ? 0x00u : 0x80u, (uint32_t)LbX, 0x040c000cu) ^ kObPair<TC>;
        const uint32_t M0 = msg_pair16<P6>(mw, 0);
        const uint32_t M = M0 & 0x0000ffffu;
        const v2s16 dx = __builtin_elementwise_sub_sat(as_v2s(L), as_v2s(M));       // [inp_X | 0]
        const uint32_t fold = sxp ^ (sxp << 16);                                      // bit 31: parity of the signs of the regular entries (the inputs other than X and Y)
        const float sigma = as_f32(0x3f800000u | (fold & 0x80000000u));
        const int mY = (int)M0 >> 24;                                                 // Y's message (upper half of the pair, << 8)
        v4f32 r;
        r.x = sigma;
        r.y = -sigma * (float)(128 + mY);
        r.z = (float)(Pm + 1);
        r.w = byte1_f32(as_u32(dx) ^ 0x8000u);                  // inp_X + 128
        rec[jj] = r;
    }
    lds_barrier();
    if (head) {
        // rows past 359 read the padding of the table and log into the padding: no per-lane predicate in the loop
        const lds_v4f_t* rp = rec + jj + B;
        lds_f32_t* lp = logv + jj + B;
        const int nsteps = (kM - 1) / B; // rows jj + k B, k = 1 .. nsteps (the last one may lie in the padding)
        __builtin_amdgcn_s_setprio(3);
        auto step = [&](const v4f32 rc) {
            *lp = c; lp += B;
            const float x = __builtin_fmaf(c, rc.x, rc.y);
            const float w = vmed3_f32(x, -rc.z, rc.z);
            const float f = w - vmed3_f32(w, -1.f, 1.f);
            c = vmed3_f32(rc.w + f, 0.f, 255.f);
        };
        // A step is six dependent instructions (~40 cycles of a lone wave), an LDS read takes 64-130: with the operands of only the
        // NEXT row in flight the walk ran at the LDS latency (~150 cycles per step, cycle stamps of round 3: 26.6 k cycles for the 179
        // steps of 3/4 normal's block-2 layer). Four rows are kept in flight; reads past the table (rows >= 360 + block) fetch
        // whatever lies there and are never used.
        v4f32 q0 = rp[0], q1 = rp[B], q2 = rp[2 * B], q3 = rp[3 * B];
        rp += 4 * B;
        int k = 0;
        for (; k + 4 <= nsteps; k += 4) {
            step(q0); q0 = rp[0];
            step(q1); q1 = rp[B];
            step(q2); q2 = rp[2 * B];
            step(q3); q3 = rp[3 * B];
            rp += 4 * B;
        }
        if (k < nsteps) { step(q0); k++; }
        if (k < nsteps) { step(q1); k++; }
        if (k < nsteps) { step(q2); k++; }
        __builtin_amdgcn_s_setprio(0);
    }
    lds_barrier();
    if (work) {
        if constexpr (!KEEP_AD) addresses();
        if (body) {
            if (!middle) LbX = lds_rd(ad[0]); // tail: the Y value its head wrote in the first phase
            pair0(LbX, (int)logv[jj] ^ (TC ? 0x80 : 0)); // (the log holds the chain's offset-binary values)
        }
        sxp ^= as_u32(d[0]);
        int m4[4] = { p0, p1, (int)(as_u32(a[0]) & 0xffffu), (int)(as_u32(a[0]) >> 16) };
        int n0, n1;
        two_smallest<4>(m4, n0, n1);
        n0 &= 0x7f00; n1 &= 0x7f00;
        const int n0m = (int)__builtin_elementwise_sub_sat((uint32_t)n0, 256u), n1m = (int)__builtin_elementwise_sub_sat((uint32_t)n1, 256u);
        const int B0 = n0, B1 = n0 + n1m - n0m, T = n1m + n0;
        const v2s16 B0p = { (short)B0, (short)B0 }, B1p = { (short)B1, (short)B1 }, Tp = { (short)T, (short)T };
        const uint32_t tm = (uint32_t)((int)(sxp ^ (sxp << 16)) >> 31);
        uint32_t R[NP];
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int j = 0; j < NP; j++) {
            const v2s16 cl = __builtin_elementwise_min(__builtin_elementwise_max(a[j], B0p), B1p);
            const v2s16 other = Tp - cl;
            const v2s16 sg = as_v2s(as_u32(d[j]) ^ tm) >> (v2s16){ 15, 15 };
            const v2s16 out = as_v2s(as_u32(other) ^ as_u32(sg)) - sg;
            const uint32_t nl = (as_u32(__builtin_elementwise_add_sat(d[j], out)) ^ kObPair<TC>) >> 8;
            if (j == 0) {
                if (jj + B >= kM) lds_wr(ad[0], (int)nl);   // a tail row is the last writer of its X bit (the others handed X down the chain)
                if (body) lds_wr_hi(ad[1], nl);             // heads wrote Y in P1 (by now a tail row may have replaced it)
            } else {
                lds_wr(ad[2 * j], (int)nl);
                if (!(ODD && j == NP - 1)) lds_wr_hi(ad[2 * j + 1], nl);
            }
            R[j] = as_u32(__builtin_elementwise_min(__builtin_elementwise_max(out, (v2s16){ -32 * 256, -32 * 256 }), (v2s16){ 31 * 256, 31 * 256 }));
        }
        __builtin_amdgcn_s_setprio(3);
        if (ODD) R[NP - 1] &= 0x0000ffffu;
        msg_pack16<P6, NP, DMAX / 4>(R, nm);
    }
}

// Hazard layer (two or more entries of one group, ldpc_schedule.h): the reference's strictly ordered update
// makes check j see what checks j' < j wrote to the bits they share. Only the NC hazard entries (placed first)
// carry that dependency, so the check node is split in three:
//   P1  all 360 rows in parallel: regular entries are read and reduced to a partial (min0, min1, signs);
//   P2  ascending blocks of B_i rows, one workgroup barrier per block: the rows of the block read their
//       hazard bits (now final with respect to all earlier rows), complete (min0, min1, signs), and write the
//       hazard bits back;
//   P3  all rows in parallel: outputs of the regular entries.
// The result is identical to the sequential order: inside a block no two rows share a bit, blocks ascend, and
// a regular entry's bits are touched by exactly one row of the layer.
// NC (2, 4 or 8) is the number of entries handled in P2: the hazard entries, rounded up with regular data entries
// (moving a regular entry into the ordered part does not change the result).
// LDS scratch of a lane chain with block size B: (360 + B) per-row records (dwords) + as many log bytes
__host__ __device__ constexpr int lane_chain_words(int block) { return (kM + block) + (kM + block + 3) / 4; }
#ifndef DVBS2_LANE_CHAIN_MAXDEG_V2P
#define DVBS2_LANE_CHAIN_MAXDEG_V2P 32 // the lane chain in hazard nodes with the packed first / last phase (their state is smaller): 9/10 normal 83.9 -> 86.3 k (round 5)
#endif
constexpr int kLaneChainMaxDegV2p = DVBS2_LANE_CHAIN_MAXDEG_V2P;
constexpr int kLaneChainMaxDeg = 28;                                  // not instantiated for the big variants nor for the
                                                                      // 80-VGPR parity-in-records kernel (registers)
constexpr int kMaxHazard = 8;     // ordered entries per check in the common builds, kMaxHazardHz2 in the HZ2 builds (ldpc_layered_kernel)
constexpr int kMaxHazardHz2 = 12;
constexpr int kMaxHazard12Dmax = 28; // (the degree class 32 has the two-level walk only: twelve ordered entries on top of 30 edges do not fit its registers)
constexpr int kHazardWalk = 15; // header code: too many hazard entries, fall back to the single-wave chunk walk
template <int DEG, int NC, bool LAYER0, bool PR = false, bool LAST = false, bool TWO = false /*two-level walk compiled in*/,
          bool LR = false /*low-register form: a regular entry keeps ONE word pm = |Lb - mb| << 8 | (inp & 0xff) between the phases and its address is computed twice (two-level-chain layers of the classes >= 24)*/,
          bool TLC = false /*two-level walk with the near pair as a LANE CHAIN (round 3), see below*/,
          bool CHAINOK = true /*false: no lane chain in this build (the 80-VGPR build since round 4, see kLaneChainBuilt)*/,
          bool CLASS8 = false /*the kernel of the degree class <= 8: early pair reads, walk on absolute addresses (DVBS2_EARLY_PAIR_MAXDEG)*/,
          bool V2P = false /*round 5: FIRST and LAST phase in the packed form of check_node_v2 (pairs of regular entries in the halves of one
                             register, one-add addresses from this wave's record, two's complement messages in pair-byte order); the ordered
                             phase in between is untouched. `ent` is then the per-wave record: S0w[DMAXV], lane masks of the first NFIXH slots*/,
          int DMAXV = 0, bool P6 = false, bool TC = false /*LLR bytes in LDS are two's complement (the builds with packed nodes)*/>
__device__ __forceinline__ void check_node_hazard(uint8_t* __restrict__ lds, const uint32_t* ent, int jj, int lb, bool work,
                                                  int block, int block2 /*two-level walk: rows per outer block, 0 = off*/, const uint32_t* mw, uint32_t* nm, int own_in, int* carry,
                                                  lds_u32_t* tab /*lane_chain_words(block) of LDS scratch when the layer is a lane chain*/,
                                                  volatile lds_i32_t* hb_ctr, int& hb_epoch, const int hb_lane /*frame barrier state*/,
                                                  unsigned long long* ph = nullptr /*timing builds: cycles per phase of this node (8 slots), else null*/)
{
    unsigned long long tph = ph ? __builtin_readcyclecounter() : 0ull;
#define DVBS2_PH(i) do { if (ph) { const unsigned long long t_ = __builtin_readcyclecounter(); ph[i] += t_ - tph; tph = t_; } } while (0)
    constexpr bool OWN_REG = PR && !LAST;     // entry DEG-2 (see check_node)
    constexpr bool PREV_REG = PR && !LAYER0;  // entry DEG-1
    static_assert(!(LR && PR), "the low-register form is for the classic layout");
    static_assert(!V2P || (!LAYER0 && !PR && !LR && (NC % 2) == 0 && DEG - NC >= 2), "packed phases: regular layers of the classic layout, ordered entries in pairs");
    // ---- packed first / last phase (V2P): state of the regular PAIRS between the phases (check_node_v2) ----
    constexpr int NP = (DEG + 1) / 2;                  // pairs; pairs 0 .. NC/2 - 1 hold the ordered entries
    constexpr int NPH = NC / 2;
    constexpr bool ODD = (DEG & 1) != 0;               // the upper half of the last pair is a pad
    constexpr int NFIXH = V2P ? ((DMAXV / 2) < DEG - 2 ? (DMAXV / 2) : DEG - 2) : 0; // fix slots of the record: the NC ordered entries first, then the mixed regular ones
    constexpr bool KEEP_AD = !V2P || DEG <= 16;        // high degrees compute the regular entries' addresses again in the last phase
    v2s16 dP[V2P ? NP : 1], aP[V2P ? NP : 1];
    uint32_t sxp = 0;
    constexpr int NAD = LR ? NC : DEG; // LR: only the ordered entries keep their addresses
    int ad[NAD], inp[LR ? NC : DEG], mg[LR ? NC : DEG];
    int pm[LR ? DEG : 1];  // LR: regular entry k keeps pm[k]
    // Lane-chain layers of the low degree classes: the pair's two LLR bytes are read WITH the regular entries (one LDS round trip for all
    // seven instead of three in a row on the wave that walks the chain afterwards). A value read here is used only by the rows for which
    // no earlier row of this layer writes that bit: entry 0 of the rows below 360 - block, entry 1 of the heads.
    constexpr bool kEarlyPair = CLASS8 && DEG <= DVBS2_EARLY_PAIR_MAXDEG && NC == 2 && !LR && !PR && CHAINOK && DEG <= DVBS2_FWALK_MAXDEG && !V2P;
    int Lh01[2] = { 0x80, 0x80 };
    int p0 = 0, p1 = 0;
    const int jjb = jj + lb, jjb360 = jjb - kM;
    int min0 = 127, min1 = 127, signs = 0;
    int spare = 0x80;
    const bool last_valid = !LAYER0 || jj != 0;
    auto addr = [&](int k) -> int {
        if (k >= DEG - 2 && !(LAYER0 && k == DEG - 1)) return jjb + (int)ent[2 * k];
        return wrap_addr(jj, jjb, jjb360, ent[2 * k], ent[2 * k + 1]);
    };
    __builtin_amdgcn_s_setprio(0); // as in check_node; the ordered steps below run at the top priority
    auto v2p_addresses = [&](int first) { // one add per entry from this wave's record (+ 360 under the record's lane mask in the fix slots)
#pragma unroll
        for (int k = 0; k < DEG; k++) if (k >= first) ad[k] = jjb + (int)ent[k];
#pragma unroll
        for (int k = 0; k < NFIXH; k++) if (k >= first) ad[k] = fix_wrap(ad[k], ent[DMAXV + 2 * k], ent[DMAXV + 2 * k + 1]);
    };
    if constexpr (V2P) {
        if (work) {
            v2p_addresses(0);
            int Lb[DEG];
#pragma unroll
            for (int k = NC; k < DEG; k++) Lb[k] = lds_rd(ad[k]);
#pragma unroll
            for (int j = NPH; j < NP; j++) {
                const uint32_t M = msg_pair16<P6>(mw, j);
                const uint32_t hi = (ODD && j == NP - 1) ? (TC ? 0x00u : 0x80u) : (uint32_t)Lb[2 * j + 1];
                const uint32_t L = __builtin_amdgcn_perm(hi, (uint32_t)Lb[2 * j], 0x040c000cu) ^ kObPair<TC>;
                dP[j] = __builtin_elementwise_sub_sat(as_v2s(L), as_v2s(M));
                sxp ^= as_u32(dP[j]);
                aP[j] = __builtin_elementwise_max(dP[j], __builtin_elementwise_sub_sat(as_v2s(0u), dP[j]));
            }
            int mgr[DEG - NC];
#pragma unroll
            for (int k = NC; k < DEG; k++) mgr[k - NC] = (k & 1) ? (int)(as_u32(aP[k >> 1]) >> 16) : (int)(as_u32(aP[k >> 1]) & 0xffffu);
            two_smallest<DEG - NC>(mgr, p0, p1); // raw |inp| << 8 of the regular entries
            min0 = (int)(__builtin_elementwise_sub_sat((uint32_t)(p0 & 0x7f00), 256u) >> 8); // R2 on the partial minimum: 0 .. 126 (what the ordered phase takes)
            signs = (int)(sxp ^ (sxp << 16));                                                  // bit 31: parity of the regular entries' signs (the only bit the ordered phase looks at)
        }
    } else
    if constexpr (LR) {
        if (work) {
#pragma unroll
            for (int k = 0; k < NC; k++) ad[k] = addr(k);
#pragma unroll
            for (int k = NC; k < DEG; k++) {
                const int Lb = lds_rdx<TC>(addr(k));
                const int mb = (int)((mw[k >> 2] >> (8 * (k & 3))) & 0xffu);
                int d = min(max(Lb - mb, -128), 127);
                const int magp = (int)__builtin_amdgcn_sad_u16((uint32_t)Lb, (uint32_t)mb, 0u);
                if (LAYER0 && k == DEG - 1) { d = last_valid ? d : 0; pm[k] = last_valid ? pm_pack(magp, d) : (kMagAbsent << 8); }
                else pm[k] = pm_pack(magp, d);
                signs ^= d;
            }
            two_smallest<DEG - NC>(pm + NC, p0, p1);
            min0 = pm_min_clamped(p0); min1 = pm_min_clamped(p1);
        }
    } else
    if (work) {
#pragma unroll
        for (int k = 0; k < DEG; k++) {
            if (k >= DEG - 2 && !(LAYER0 && k == DEG - 1)) ad[k] = jjb + (int)ent[2 * k];
            else ad[k] = wrap_addr(jj, jjb, jjb360, ent[2 * k], ent[2 * k + 1]);
        }
#pragma unroll
        for (int k = 0; k < DEG; k++) {
            if (kEarlyPair && k < 2) Lh01[k] = lds_rdx<TC>(ad[k]); // (see the lane chain below: issued with the regular reads, used only where still valid)
            if (k >= NC) { // regular entry
                const int Lb = (OWN_REG && k == DEG - 2) ? own_in : (PREV_REG && k == DEG - 1) ? *carry : lds_rdx<TC>(ad[k]);
                const int mb = (int)((mw[k >> 2] >> (8 * (k & 3))) & 0xffu);
                int d = min(max(Lb - mb, -128), 127);
                int mag = mag_raw(Lb, mb);
                if (LAYER0 && k == DEG - 1) { d = last_valid ? d : 0; mag = last_valid ? mag : kMagAbsent; }
                inp[k] = d; mg[k] = mag;
                signs ^= d;
            }
        }
        two_smallest<DEG - NC>(mg + NC, min0, min1); // raw magnitudes of the regular entries (see mag_raw)
        min0 = clamp_mag(min0); min1 = clamp_mag(min1);
    }
    DVBS2_PH(0); // P1: regular entries read and reduced
    if constexpr (!V2P) {
#pragma unroll
    for (int w = 0; w < (DEG + 3) / 4; w++) nm[w] = 0;
    }
    // P2 keeps only what the NEXT block needs on its critical path: the new hazard LLRs. For hazard entry k the
    // magnitude sent back is the minimum over all OTHER entries = min(partial min0 of the regular entries, the
    // other hazard magnitudes) and the sign is the xor of all other signs; the merge of the hazard entries into
    // (min0, min1, signs) for P3 and the hazard message bytes are computed after the loop.
    if (PR && (DEG + 3) / 4 < 2) nm[1] = 0;
    int hout[NC], hmb[NC];
#pragma unroll
    for (int k = 0; k < NC; k++) {
        hout[k] = 0; inp[k] = 0; mg[k] = 127;
        if constexpr (V2P) hmb[k] = (int)(((mw[k >> 2] >> (8 * (((k >> 1) & 1) + 2 * (k & 1)))) & 0xffu) ^ 0x80u); // pair-byte order, two's complement -> offset binary
        else hmb[k] = (int)((mw[k >> 2] >> (8 * (k & 3))) & 0xffu);
    }
    // One ordered step per block of `block` rows. A step is a chain of dependent instructions of a single wave (the
    // next block reads what this one wrote), so its length is what a hazard layer costs: rel = jj - start is kept
    // incrementally (one subtract + one unsigned compare select the rows of the block).
    __builtin_amdgcn_s_setprio(3);
    bool lane_chain = false;
    // (the low-register form has room for it at every degree)
    constexpr bool kLaneChainBuilt = NC == 2 && (LR || DEG <= (V2P ? kLaneChainMaxDegV2p : kLaneChainMaxDeg)) && !PR && CHAINOK;
    if constexpr (kLaneChainBuilt) lane_chain = tab != nullptr; // wave-uniform (header bit 12)
    if constexpr (kLaneChainBuilt) if (lane_chain) {
        // LANE CHAIN (one hazard pair, block <= 128, host-ordered so that entry 0's bit of row r is entry 1's bit of
        // row r + block). A lone wave issues one instruction per 4-7 cycles whatever it is, so an ordered step costs
        // its instruction count: the recurrence r -> r + block is walked by the `block` lanes that own rows
        // 0..block-1 with the chained LLR in a register, ~20 instructions per step, no exec-mask bookkeeping, no LDS
        // hand-over, no barrier per step; everything else happens before and after, in parallel over all rows. Round 3: two
        // barriers per layer instead of four (cycle stamps, DVBS2_PH: the heads' step, the publishing pass and their barriers
        // cost a chain layer of table B4 ~1.4 k of its ~5 k cycles):
        //   A  together with the first phase (no barrier in between): rows < 360 - block read entry 0 -- no earlier row of
        //      this layer writes that bit -- ; the heads (rows < block, nothing precedes them) do their full two-entry step and
        //      write entry 1 at once (the only reader of that bit is the tail row r + 360 - block, after the walk); the other
        //      rows publish {inp0, partial min0, partial sign, message byte 1}                             | barrier
        //   C  chain lanes: incoming entry-1 LLR -> new entry-0 LLR of row r, incoming value logged      | barrier
        //   D  rows >= block: (tails first read entry 0 = what their head wrote in A) complete both outputs from the
        //      logged value; write entry 1; the last row of a chain also writes entry 0 (in the reference's order it is the
        //      final writer of that bit). No barrier towards the outputs of the regular entries: other bits.
        // (The walk on exact small integers in float -- six instructions per step as in the packed chain node instead of ~20 -- was
        // measured here too in round 3: the 16-byte operand records and the float state cost every build of every degree class 4-6
        // VGPRs; B4 113.2 k -> 112.4 k, the 80-VGPR and one-frame builds -4 ... -8 %. It stays in the packed chain node.)
        lds_byte_t* ulog = reinterpret_cast<lds_byte_t*>(tab + kM + block); // after the per-row records (360 rows + one block of padding)
        int chained = 0x80;
        // Round 4, degree class 8 (DVBS2_FWALK): the walk on exact small integers in float with the operands of FOUR rows in flight. The
        // integer step is ~17 dependent VALU instructions and one record read ahead: a lone wave needs ~110 cycles per row either way
        // (issue ~4 cycles per instruction, an LDS read 100-130), 1.4 k cycles for the 8-10 rows of table B4's chains. In float the step is
        // six instructions (fma, two med3 with a negated operand, sub, add, clamp -- the packed chain node's step, check_node_chain_v2)
        // and with four 16-byte records in flight the LDS latency is covered.
        //   record of row r: { sigma, -sigma m1, P + 1, inp0 + 128 }  (sigma = +-1: partial sign; m1: entry 1's message, offset binary;
        //   P: partial minimum); incoming entry-1 LLR c (offset binary):  x = sigma (c - m1), w = clamp(x, -(P+1), P+1),
        //   out = w - sgn(w) = sgn(x) min(P, max(|x| - 1, 0)),  c' = clamp(inp0 + 128 + out, 0, 255)
        constexpr bool kFloatWalk = (DVBS2_FWALK != 0) && DEG <= DVBS2_FWALK_MAXDEG && !LR && !PR;
        lds_v4f_t* frec = lds_align16<lds_v4f_t>(tab);                                                   // [360 + block]
        lds_f32_t* flog = reinterpret_cast<lds_f32_t*>(frec) + 4 * (kM + kChainMaxBlock);              // [360 + block]
        auto publish = [&]() {
            if constexpr (kFloatWalk) {
                const float sigma = as_f32(0x3f800000u | ((uint32_t)signs & 0x80000000u));
                v4f32 r;
                r.x = sigma; r.y = -sigma * (float)hmb[1]; r.z = (float)(min0 + 1); r.w = (float)(inp[0] + 128);
                frec[jj] = r;
            } else
            tab[jj] = ((uint32_t)inp[0] & 0x1ffu) | ((uint32_t)min0 << 9) | (((uint32_t)signs >> 31) << 16) | ((uint32_t)hmb[1] << 24);
        };
        const bool head = work && jj < block, body = work && jj >= block;
        // (degrees above 20 without the low-register form keep the round-2 order -- heads | barrier | publishing | barrier | walk |
        // barrier | completion | barrier --: reading entry 0 inside the first phase costs the degree class 28 sixteen more spilled
        // registers and table B10 7 %)
#ifndef DVBS2_TWO_BARRIER_V2P
#define DVBS2_TWO_BARRIER_V2P 0 // experiments (round 6): the two-barrier order also in the packed hazard nodes of degree > 20
#endif
        constexpr bool kTwoBarrier = LR || DEG <= 20 || (V2P && DVBS2_TWO_BARRIER_V2P != 0);
        const bool orig0 = work && (kTwoBarrier ? jj + block < kM : jj < block); // entry 0 still holds its value from before the layer (every head is one: block <= 128)
        if (orig0) {
            const int L0 = kEarlyPair ? Lh01[0] : lds_rdx<TC>(ad[0]);
            inp[0] = min(max(L0 - hmb[0], -128), 127);
            mg[0] = mag_raw(L0, hmb[0]);
        }
        if (head) {
            const int L1 = kEarlyPair ? Lh01[1] : lds_rdx<TC>(ad[1]);
            inp[1] = min(max(L1 - hmb[1], -128), 127);
            mg[1] = mag_raw(L1, hmb[1]);
            int o0, o1;
            asm("v_med3_i32 %0, %1, 0, %2" : "=v"(o0) : "v"(mg[1]), "v"(min0));
            asm("v_med3_i32 %0, %1, 0, %2" : "=v"(o1) : "v"(mg[0]), "v"(min0));
            const int s0 = (signs ^ inp[1]) >> 31, s1 = (signs ^ inp[0]) >> 31;
            hout[0] = (o0 ^ s0) - s0;
            hout[1] = (o1 ^ s1) - s1;
            chained = sat_sum_u8(inp[0], hout[0]);
            lds_wrx<TC>(ad[1], sat_sum_u8(inp[1], hout[1]));
        } else if (kTwoBarrier && orig0)
            publish();
        if constexpr (!kTwoBarrier) {
            lds_barrier();
            if (body) { // every row below the heads, tails included (their entry 0 is what a head just wrote)
                const int L0 = lds_rdx<TC>(ad[0]);
                inp[0] = min(max(L0 - hmb[0], -128), 127);
                mg[0] = mag_raw(L0, hmb[0]);
                publish();
            }
        }
        DVBS2_PH(1); // chain heads + publishing
        lds_barrier();
        DVBS2_PH(3); // barrier
        if constexpr (kFloatWalk && CLASS8 && DEG <= DVBS2_WALK_ABS_MAXDEG) { if (head) {
            // absolute LDS addresses, ONE running address for the records and one for the log (through typed pointers the compiler kept
            // eight offsets and added the array base at every access: four address instructions per step of a lone wave)
            int ra = (int)(uint32_t)(size_t)(frec + jj + block), la = (int)(uint32_t)(size_t)(flog + jj + block);
            const int rs = block * 16, ls = block * 4;
            auto ld = [](int a) -> v4f32 { return *reinterpret_cast<const lds_v4f_t*>((size_t)(uint32_t)a); };
            const int nsteps = (kM - 1) / block;
            float c = (float)chained;
            auto step = [&](const v4f32 rc) {
                *reinterpret_cast<lds_f32_t*>((size_t)(uint32_t)la) = c; la += ls;
                const float x = __builtin_fmaf(c, rc.x, rc.y);
                const float w = vmed3_f32(x, -rc.z, rc.z);
                const float f = w - vmed3_f32(w, -1.f, 1.f);
                c = vmed3_f32(rc.w + f, 0.f, 255.f);
            };
            v4f32 q0 = ld(ra), q1 = ld(ra + rs), q2 = ld(ra + 2 * rs), q3 = ld(ra + 3 * rs);
            ra += 4 * rs;
            int k = 0;
            for (; k + 4 <= nsteps; k += 4) {
                step(q0); q0 = ld(ra); ra += rs;
                step(q1); q1 = ld(ra); ra += rs;
                step(q2); q2 = ld(ra); ra += rs;
                step(q3); q3 = ld(ra); ra += rs;
            }
            if (k < nsteps) { step(q0); k++; }
            if (k < nsteps) { step(q1); k++; }
            if (k < nsteps) { step(q2); k++; }
        } } else
        if constexpr (kFloatWalk) { if (head) {
            const lds_v4f_t* rp = frec + jj + block;
            lds_f32_t* lp = flog + jj + block;
            const int nsteps = (kM - 1) / block; // rows jj + k block, k = 1 .. nsteps (the last one may lie in the padding)
            float c = (float)chained;
            auto step = [&](const v4f32 rc) {
                *lp = c; lp += block;
                const float x = __builtin_fmaf(c, rc.x, rc.y);
                const float w = vmed3_f32(x, -rc.z, rc.z);
                const float f = w - vmed3_f32(w, -1.f, 1.f);
                c = vmed3_f32(rc.w + f, 0.f, 255.f);
            };
            v4f32 q0 = rp[0], q1 = rp[block], q2 = rp[2 * block], q3 = rp[3 * block]; // (reads past the table fetch scratch that is never used)
            rp += 4 * block;
            int k = 0;
            for (; k + 4 <= nsteps; k += 4) {
                step(q0); q0 = rp[0];
                step(q1); q1 = rp[block];
                step(q2); q2 = rp[2 * block];
                step(q3); q3 = rp[3 * block];
                rp += 4 * block;
            }
            if (k < nsteps) { step(q0); k++; }
            if (k < nsteps) { step(q1); k++; }
            if (k < nsteps) { step(q2); k++; }
        } } else
        if (head) {
            // rows past 359 read the padding of the table and log into the padding: no per-lane predicate in the loop; the record
            // of a tail row (last of its chain) is not written: what is computed from it is never used
            const lds_u32_t* tp = tab + jj + block;
            lds_byte_t* up = ulog + jj + block;
            uint32_t t = *tp;
            for (int first = block; first < kM; first += block) {
                const uint32_t tc = t;
                tp += block;
                t = *tp; // next row's record, in flight during this step
                const int i0 = (int)(tc << 23) >> 23, P = (int)((tc >> 9) & 0x7fu), m1 = (int)(tc >> 24);
                const int sw = (int)(tc << 15); // partial sign in bit 31
                *up = (uint8_t)chained; up += block;
                const int i1 = min(max(chained - m1, -128), 127);
                const int g1 = mag_raw(chained, m1);
                int o0;
                asm("v_med3_i32 %0, %1, 0, %2" : "=v"(o0) : "v"(g1), "v"(P));
                const int s0 = (sw ^ i1) >> 31;
                chained = sat_sum_u8(i0, (o0 ^ s0) - s0);
            }
        }
        DVBS2_PH(4); // walk
        lds_barrier();
        DVBS2_PH(5); // barrier after the walk
        if (body) {
            if (kTwoBarrier && !orig0) { // tail: entry 0 = the entry-1 value its head wrote before the walk
                const int L0 = lds_rdx<TC>(ad[0]);
                inp[0] = min(max(L0 - hmb[0], -128), 127);
                mg[0] = mag_raw(L0, hmb[0]);
            }
            const int L1 = kFloatWalk ? (int)flog[jj] : (int)ulog[jj];
            inp[1] = min(max(L1 - hmb[1], -128), 127);
            mg[1] = mag_raw(L1, hmb[1]);
            int o0, o1;
            asm("v_med3_i32 %0, %1, 0, %2" : "=v"(o0) : "v"(mg[1]), "v"(min0));
            asm("v_med3_i32 %0, %1, 0, %2" : "=v"(o1) : "v"(mg[0]), "v"(min0));
            const int s0 = (signs ^ inp[1]) >> 31, s1 = (signs ^ inp[0]) >> 31;
            hout[0] = (o0 ^ s0) - s0;
            hout[1] = (o1 ^ s1) - s1;
            lds_wrx<TC>(ad[1], sat_sum_u8(inp[1], hout[1]));
            if (jj + block >= kM) lds_wrx<TC>(ad[0], sat_sum_u8(inp[0], hout[0]));
        }
    }
    // TWO-LEVEL LANE CHAIN (NC >= 4, block2 > 0, header bit 12; degree class 32 without the heavy-hazard paths). As in the two-level
    // walk below, ONE pair of ordered entries (0, 1; host-ordered: entry 0's bit of row r is entry 1's bit of row r + block) is closer
    // than every other pair (>= block2 >= 2 block rows apart), and the rows are processed in outer blocks of block2 rows. Inside an
    // outer block only the near pair is sequential -- and it is walked like a single-pair lane chain: the `block` lanes that own rows
    // 0 .. block-1 carry the chained LLR in a register from row to row ACROSS the outer blocks, ~20 instructions per row, no LDS
    // hand-over. Per outer block:
    //   a  its rows read their far entries (final: the rows that share those bits lie in other outer blocks), fold them into the
    //      partial minimum / sign, read entry 0 (untouched so far, or -- last `block` rows -- what a head wrote in the first outer
    //      block) and publish {inp0, partial min, partial sign, message byte 1}; heads (first outer block) do their two-entry step | barrier
    //   b  the chain lanes walk the rows of this outer block, logging what arrives at each row                               | barrier
    //   c  its rows complete the near pair from the logged value, then the far entries (minimum over all other entries), and
    //      write those LLRs                                                                                                  | barrier
    // 9/10 normal, layer 5 (pairs 4, 53, 84, 86 rows apart): 90 eight-entry steps through LDS (~80 k cycles, a fifth of the sweep)
    // become 7 outer blocks + 89 register steps.
    constexpr bool kTlcBuilt = TLC && (NC == 4 || NC == 8) && !PR;
    bool tlc = false;
    if constexpr (kTlcBuilt) tlc = block2 > 0 && tab != nullptr; // wave-uniform
    if constexpr (kTlcBuilt) if (tlc) {
        lds_byte_t* ulog = reinterpret_cast<lds_byte_t*>(tab + kM + block);
        constexpr bool kTlcFloat = V2P && DMAXV >= DVBS2_TLC_FWALK_MIN_DMAX;
        lds_v4f_t* trec = lds_align16<lds_v4f_t>(tab);                                      // kTlcFloat: [360 + block] operand records { sigma, -sigma m1, P + 1, inp0 + 128 }
        lds_f32_t* tlog = reinterpret_cast<lds_f32_t*>(trec) + 4 * (kM + kChainMaxBlock); //            [360 + block] the value that arrived at a row
        int chained = 0x80;
        float cf = 0.f;
        const bool head = work && jj < block;
        int rnext = jj + block; // chain lanes: the next row to visit
        for (int sb = 0; sb < kM; sb += block2) {
            const int sb_end = min(sb + block2, kM);
            const bool in_sb = work && (uint32_t)(jj - sb) < (uint32_t)block2;
            int minF = min0, signsF = signs;
            if (in_sb) {
                int Lh[NC];
#pragma unroll
                for (int k = 2; k < NC; k++) Lh[k] = lds_rdx<TC>(ad[k]);
                const int L0 = lds_rdx<TC>(ad[0]);
#pragma unroll
                for (int k = 2; k < NC; k++) {
                    inp[k] = min(max(Lh[k] - hmb[k], -128), 127);
                    mg[k] = mag_offset(Lh[k], hmb[k]);
                    signsF ^= inp[k];
                    minF = min(minF, mg[k]);
                }
                inp[0] = min(max(L0 - hmb[0], -128), 127);
                mg[0] = mag_raw(L0, hmb[0]);
                if (head) { // (first outer block: block2 >= 2 block)
                    const int L1 = lds_rdx<TC>(ad[1]);
                    inp[1] = min(max(L1 - hmb[1], -128), 127);
                    mg[1] = mag_raw(L1, hmb[1]);
                    int o0, o1;
                    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(o0) : "v"(mg[1]), "v"(minF));
                    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(o1) : "v"(mg[0]), "v"(minF));
                    const int s0 = (signsF ^ inp[1]) >> 31, s1 = (signsF ^ inp[0]) >> 31;
                    hout[0] = (o0 ^ s0) - s0;
                    hout[1] = (o1 ^ s1) - s1;
                    chained = sat_sum_u8(inp[0], hout[0]);
                    cf = (float)chained;
                    lds_wrx<TC>(ad[1], sat_sum_u8(inp[1], hout[1]));
                } else if constexpr (kTlcFloat) {
                    const float sigma = as_f32(0x3f800000u | ((uint32_t)signsF & 0x80000000u));
                    v4f32 r;
                    r.x = sigma; r.y = -sigma * (float)hmb[1]; r.z = (float)(minF + 1); r.w = (float)(inp[0] + 128);
                    trec[jj] = r;
                } else
                    tab[jj] = ((uint32_t)inp[0] & 0x1ffu) | ((uint32_t)minF << 9) | (((uint32_t)signsF >> 31) << 16) | ((uint32_t)hmb[1] << 24);
            }
            lds_barrier();
            if constexpr (kTlcFloat) { if (head && rnext < sb_end) {
                // (as the single-pair chains: x = sigma (c - m1), w = clamp(x, -(P+1), P+1), out = w - sgn(w), c' = clamp(inp0 + 128 + out, 0, 255))
                v4f32 q = trec[rnext];
                for (; rnext < sb_end; rnext += block) {
                    const v4f32 rc = q;
                    q = trec[rnext + block]; // next row's record (valid when that row belongs to this outer block; reloaded otherwise)
                    tlog[rnext] = cf;
                    const float x = __builtin_fmaf(cf, rc.x, rc.y);
                    const float w = vmed3_f32(x, -rc.z, rc.z);
                    const float f = w - vmed3_f32(w, -1.f, 1.f);
                    cf = vmed3_f32(rc.w + f, 0.f, 255.f);
                }
            } } else
            if (head && rnext < sb_end) {
                uint32_t t = tab[rnext];
                for (; rnext < sb_end; rnext += block) {
                    const uint32_t tc = t;
                    t = tab[rnext + block]; // next row's record (valid when that row belongs to this outer block; reloaded otherwise)
                    const int i0 = (int)(tc << 23) >> 23, P = (int)((tc >> 9) & 0x7fu), m1 = (int)(tc >> 24);
                    const int sw = (int)(tc << 15);
                    ulog[rnext] = (uint8_t)chained;
                    const int i1 = min(max(chained - m1, -128), 127);
                    const int g1 = mag_raw(chained, m1);
                    int o0;
                    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(o0) : "v"(g1), "v"(P));
                    const int s0 = (sw ^ i1) >> 31;
                    chained = sat_sum_u8(i0, (o0 ^ s0) - s0);
                }
            }
            lds_barrier();
            if (in_sb) {
                if (!head) {
                    const int L1 = kTlcFloat ? (int)tlog[jj] : (int)ulog[jj];
                    inp[1] = min(max(L1 - hmb[1], -128), 127);
                    mg[1] = mag_raw(L1, hmb[1]);
                    int o0, o1;
                    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(o0) : "v"(mg[1]), "v"(minF));
                    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(o1) : "v"(mg[0]), "v"(minF));
                    const int s0 = (signsF ^ inp[1]) >> 31, s1 = (signsF ^ inp[0]) >> 31;
                    hout[0] = (o0 ^ s0) - s0;
                    hout[1] = (o1 ^ s1) - s1;
                    lds_wrx<TC>(ad[1], sat_sum_u8(inp[1], hout[1]));
                    if (jj + block >= kM) lds_wrx<TC>(ad[0], sat_sum_u8(inp[0], hout[0])); // last row of its chain: final writer of that bit
                }
                mg[0] = clamp_mag(mg[0]); mg[1] = clamp_mag(mg[1]); // (raw above; everything below and after the loop takes clamped ones)
                const int xall = signsF ^ inp[0] ^ inp[1];
                int pre[NC + 1], suf[NC + 1];
                pre[0] = min0; suf[NC] = 127;
#pragma unroll
                for (int k = 0; k < NC; k++) pre[k + 1] = min(pre[k], mg[k]);
#pragma unroll
                for (int k = NC - 1; k >= 0; k--) suf[k] = min(suf[k + 1], mg[k]);
#pragma unroll
                for (int k = 2; k < NC; k++) {
                    const int other = min(pre[k], suf[k + 1]);
                    const int sg = (xall ^ inp[k]) >> 31;
                    const int out = (other ^ sg) - sg;
                    hout[k] = out;
                    lds_wrx<TC>(ad[k], sat_sum_u8(inp[k], out));
                }
            }
            if (sb + block2 < kM) lds_barrier(); // the next outer block reads what this one wrote
        }
    }
    // TWO-LEVEL WALK (NC >= 4, block2 > 0). The block size B is the distance of the NEAREST hazard pair only; every other pair
    // of hazard entries is at least block2 >= 2 B rows apart. So the rows are walked in outer blocks of block2 rows: at its start the
    // rows of an outer block read their FAR hazard entries (2 .. NC-1: final with respect to all earlier outer blocks, untouched
    // inside this one) and fold them into the partial result; then only the near pair (entries 0, 1; host-ordered) goes through the
    // ordered steps of B rows -- a two-entry step instead of an NC-entry one -- and at the end of the outer block its rows write the
    // far entries back. 360 / B short steps + 360 / block2 long ones instead of 360 / B long ones (B11 layer 5: B = 4).
    bool two_level = false;
    // (degree 29 and above: only the eight-entry form is compiled -- every instantiation costs that class registers, and 9/10 normal,
    // the table it is there for, has its block-4 layer with eight ordered entries)
    constexpr bool kTwoBuilt = TWO && NC >= 4 && (DEG < 29 || NC == 8);
    if constexpr (kTwoBuilt) two_level = block2 > 0;
    if constexpr (kTwoBuilt) if (two_level) {
        for (int sb = 0; sb < kM; sb += block2) {
            const bool in_sb = work && (uint32_t)(jj - sb) < (uint32_t)block2;
            int minF = min0, signsF = signs;
            if (in_sb) {
                int Lh[NC];
#pragma unroll
                for (int k = 2; k < NC; k++) Lh[k] = lds_rdx<TC>(ad[k]);
#pragma unroll
                for (int k = 2; k < NC; k++) {
                    inp[k] = min(max(Lh[k] - hmb[k], -128), 127);
                    mg[k] = mag_offset(Lh[k], hmb[k]);
                    signsF ^= inp[k];
                    minF = min(minF, mg[k]);
                }
            }
            const int sb_end = min(sb + block2, kM);
            int rel = in_sb ? jj - sb : 0x40000000;
            for (int start = sb; start < sb_end; start += block, rel -= block) {
                if ((uint32_t)rel < (uint32_t)block) {
                    const int L0 = lds_rdx<TC>(ad[0]), L1 = lds_rdx<TC>(ad[1]);
                    inp[0] = min(max(L0 - hmb[0], -128), 127);
                    inp[1] = min(max(L1 - hmb[1], -128), 127);
                    mg[0] = mag_offset(L0, hmb[0]);
                    mg[1] = mag_offset(L1, hmb[1]);
                    const int o0 = min(mg[1], minF), o1 = min(mg[0], minF);
                    const int s0 = (signsF ^ inp[1]) >> 31, s1 = (signsF ^ inp[0]) >> 31;
                    hout[0] = (o0 ^ s0) - s0;
                    hout[1] = (o1 ^ s1) - s1;
                    lds_wrx<TC>(ad[0], sat_sum_u8(inp[0], hout[0]));
                    lds_wrx<TC>(ad[1], sat_sum_u8(inp[1], hout[1]));
                }
                if (start + block < sb_end && (start >> 6) != ((start + 2 * block - 1) >> 6)) lds_barrier();
            }
            if (in_sb) {
                const int xall = signsF ^ inp[0] ^ inp[1];
                int pre[NC + 1], suf[NC + 1];
                pre[0] = min0; suf[NC] = 127;
#pragma unroll
                for (int k = 0; k < NC; k++) pre[k + 1] = min(pre[k], mg[k]);
#pragma unroll
                for (int k = NC - 1; k >= 0; k--) suf[k] = min(suf[k + 1], mg[k]);
#pragma unroll
                for (int k = 2; k < NC; k++) {
                    const int other = min(pre[k], suf[k + 1]);
                    const int sg = (xall ^ inp[k]) >> 31;
                    const int out = (other ^ sg) - sg;
                    hout[k] = out;
                    lds_wrx<TC>(ad[k], sat_sum_u8(inp[k], out));
                }
            }
            // the next outer block reads what this one wrote: a barrier, unless both sit inside one and the same wavefront
            if ((sb >> 6) != ((sb + 2 * block2 - 1) >> 6)) lds_barrier();
        }
    }
    // (Round 4 measured the alternative to a workgroup barrier per block -- each wave takes only the steps of the blocks that hold its
    // rows, waits for a progress counter in LDS and goes on to its regular outputs while later waves still step: bit-exact and 4-11 %
    // SLOWER (9/10 normal -4 %, 3/5 -8 %, 8/9 -9 %, 2/3 -11 %): a hand-over through an LDS word costs ~300 cycles against ~30-50 for
    // s_barrier, more than the overlapped outputs give back. notes/r04_experiments.md.)
    int rel = (work && !lane_chain && !two_level && !tlc) ? jj : 0x40000000;
    for (int start = (lane_chain || two_level || tlc) ? kM : 0; start < kM; start += block, rel -= block) {
        if ((uint32_t)rel < (uint32_t)block) {
            if constexpr (NC == 2) {
                // two hazard entries: each one's magnitude sent back is min(partial min0, the other's magnitude) =
                // med3(raw other, 0, min0) (min0 is already clamped to [0, 126])
                const int L0 = lds_rdx<TC>(ad[0]), L1 = lds_rdx<TC>(ad[1]);
                inp[0] = min(max(L0 - hmb[0], -128), 127);
                inp[1] = min(max(L1 - hmb[1], -128), 127);
                mg[0] = mag_raw(L0, hmb[0]);
                mg[1] = mag_raw(L1, hmb[1]);
                int o0, o1;
                asm("v_med3_i32 %0, %1, 0, %2" : "=v"(o0) : "v"(mg[1]), "v"(min0));
                asm("v_med3_i32 %0, %1, 0, %2" : "=v"(o1) : "v"(mg[0]), "v"(min0));
                const int s0 = (signs ^ inp[1]) >> 31, s1 = (signs ^ inp[0]) >> 31;
                hout[0] = (o0 ^ s0) - s0;
                hout[1] = (o1 ^ s1) - s1;
                lds_wrx<TC>(ad[0], sat_sum_u8(inp[0], hout[0]));
                lds_wrx<TC>(ad[1], sat_sum_u8(inp[1], hout[1]));
            } else {
            int Lh[NC];
#pragma unroll
            for (int k = 0; k < NC; k++) Lh[k] = lds_rdx<TC>(ad[k]);
            int xall = signs;
#pragma unroll
            for (int k = 0; k < NC; k++) {
                inp[k] = min(max(Lh[k] - hmb[k], -128), 127);
                mg[k] = mag_offset(Lh[k], hmb[k]);
                xall ^= inp[k];
            }
            int pre[NC + 1], suf[NC + 1]; // pre[k] = min(min0, mg[0..k)), suf[k] = min(mg[k..NC))
            pre[0] = min0; suf[NC] = 127;
#pragma unroll
            for (int k = 0; k < NC; k++) pre[k + 1] = min(pre[k], mg[k]);
#pragma unroll
            for (int k = NC - 1; k >= 0; k--) suf[k] = min(suf[k + 1], mg[k]);
#pragma unroll
            for (int k = 0; k < NC; k++) {
                const int other = min(pre[k], suf[k + 1]);
                const int sg = (xall ^ inp[k]) >> 31;
                const int out = (other ^ sg) - sg;
                hout[k] = out;
                lds_wrx<TC>(ad[k], sat_sum_u8(inp[k], out));
            }
            }
        }
        // the next block reads what this one wrote: a workgroup barrier, unless both blocks sit inside one and the
        // same wavefront (LDS operations of a wave execute in program order)
        if ((start >> 6) != ((start + 2 * block - 1) >> 6)) lds_barrier();
    }
    if (!lane_chain || !(LR || DEG <= 20 || (V2P && DVBS2_TWO_BARRIER_V2P != 0))) lds_barrier(); // (uniform; the last phase of a two-barrier lane chain and the outputs below touch different bits)
    DVBS2_PH(6); // ordered steps of the block scheme + closing barrier / completion of the chain rows
    if constexpr (V2P) {
        // LAST PHASE, packed: the ordered entries enter the packed domain as pairs [inp << 8] (what they read in their step is final), the
        // two smallest magnitudes are merged over everything, and every pair's outputs follow as in check_node_v2. The ordered entries'
        // LLRs were written in their step (a later row may have replaced them since): only their MESSAGES are produced here -- the same
        // "minimum and sign product over all other entries" the step computed, so the two agree by construction.
        if (work) {
            if constexpr (!KEEP_AD) v2p_addresses(NC);
            int m4[NC + 2];
            m4[0] = p0; m4[1] = p1;
#pragma unroll
            for (int j = 0; j < NPH; j++) {
                const uint32_t dh = __builtin_amdgcn_perm((uint32_t)inp[2 * j + 1], (uint32_t)inp[2 * j], 0x040c000cu); // [inp_hi << 8 | inp_lo << 8]
                dP[j] = as_v2s(dh);
                sxp ^= dh;
                aP[j] = __builtin_elementwise_max(dP[j], __builtin_elementwise_sub_sat(as_v2s(0u), dP[j]));
                m4[2 + 2 * j] = (int)(as_u32(aP[j]) & 0xffffu); m4[3 + 2 * j] = (int)(as_u32(aP[j]) >> 16);
            }
            int n0, n1;
            two_smallest<NC + 2>(m4, n0, n1);
            n0 &= 0x7f00; n1 &= 0x7f00;
            const int n0m = (int)__builtin_elementwise_sub_sat((uint32_t)n0, 256u), n1m = (int)__builtin_elementwise_sub_sat((uint32_t)n1, 256u);
            const int B0 = n0, B1 = n0 + n1m - n0m, T = n1m + n0;
            const v2s16 B0p = { (short)B0, (short)B0 }, B1p = { (short)B1, (short)B1 }, Tp = { (short)T, (short)T };
            const uint32_t tm = (uint32_t)((int)(sxp ^ (sxp << 16)) >> 31);
            uint32_t R[NP];
            __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int j = 0; j < NP; j++) {
                const v2s16 cl = __builtin_elementwise_min(__builtin_elementwise_max(aP[j], B0p), B1p);
                const v2s16 other = Tp - cl;
                const v2s16 sg = as_v2s(as_u32(dP[j]) ^ tm) >> (v2s16){ 15, 15 };
                const v2s16 out = as_v2s(as_u32(other) ^ as_u32(sg)) - sg;
                if (j >= NPH) {
                    const uint32_t nl = (as_u32(__builtin_elementwise_add_sat(dP[j], out)) ^ kObPair<TC>) >> 8;
                    lds_wr(ad[2 * j], (int)nl);
                    if (!(ODD && j == NP - 1)) lds_wr_hi(ad[2 * j + 1], nl);
                }
                R[j] = as_u32(__builtin_elementwise_min(__builtin_elementwise_max(out, (v2s16){ -32 * 256, -32 * 256 }), (v2s16){ 31 * 256, 31 * 256 }));
            }
            __builtin_amdgcn_s_setprio(3);
            if (ODD) R[NP - 1] &= 0x0000ffffu;
            msg_pack16<P6, NP, DMAXV / 4>(R, nm);
        }
        DVBS2_PH(7);
        return;
    }
    if constexpr (NC == 2) { mg[0] = clamp_mag(mg[0]); mg[1] = clamp_mag(mg[1]); } // raw in the loop (127 where no step ran: idle rows)
#pragma unroll
    for (int k = 0; k < NC; k++) {
        min1 = min(max(mg[k], min0), min1);
        min0 = min(min0, mg[k]);
        signs ^= inp[k];
        nm[k >> 2] |= (uint32_t)(min(max(hout[k], -32), 31) + 128) << (8 * (k & 3));
    }
    const int s01 = min0 + min1;
    if constexpr (LR) {
        // The regular entries only know the two smallest of THEMSELVES (p0, p1); after the ordered entries were merged a regular
        // entry's "minimum of the others" is: the entry that holds p0 takes min(second regular minimum, smallest ordered magnitude),
        // every other one min(first regular minimum, smallest ordered magnitude) -- i.e. min1 for the holder of the overall
        // minimum and min0 otherwise, decided on the merged (min0, min1) like this:
        //   hmin = smallest ordered magnitude (clamped); r0, r1 = the regular minima (clamped)
        //   holder of p0:  min(r1, hmin);   others:  min(r0, hmin)
        if (work) {
            int hmin = 127;
#pragma unroll
            for (int k = 0; k < NC; k++) hmin = min(hmin, mg[k]);
            const int o_first = min(pm_min_clamped(p1), hmin), o_rest = min(pm_min_clamped(p0), hmin);
#pragma unroll
            for (int k = NC; k < DEG; k++) {
                const int other = pm[k] == p0 ? o_first : o_rest;
                const int ip = pm_inp(pm[k]);
                const int sg = (signs ^ ip) >> 31;
                const int out = (other ^ sg) - sg;
                const int nl = sat_sum_u8(ip, out);
                if (!(LAYER0 && k == DEG - 1) || last_valid) lds_wrx<TC>(addr(k), nl);
                nm[k >> 2] |= (uint32_t)(min(max(out, -32), 31) + 128) << (8 * (k & 3));
            }
        }
    } else
    if (work) {
#pragma unroll
        for (int k = 0; k < DEG; k++) {
            if (k >= NC) {
                const int other = s01 - vmed3_i32(mg[k], min0, min1); // regular entries hold raw magnitudes
                const int sg = (signs ^ inp[k]) >> 31;
                const int out = (other ^ sg) - sg;
                const int nl = sat_sum_u8(inp[k], out);
                if (OWN_REG && k == DEG - 2) *carry = nl;
                else if (PREV_REG && k == DEG - 1) spare = nl;
                else if (!(LAYER0 && k == DEG - 1) || last_valid) lds_wrx<TC>(ad[k], nl);
                nm[k >> 2] |= (uint32_t)(min(max(out, -32), 31) + 128) << (8 * (k & 3));
            }
        }
        if (PR) nm[1] = (nm[1] & 0x00ffffffu) | ((uint32_t)spare << 24);
    }
    DVBS2_PH(7); // merge + outputs of the regular entries
#undef DVBS2_PH
}

// degrees DMAX-7 .. DMAX are instantiated for kernel variant DMAX
#define DVBS2_DEG_CASE(D) case D: if constexpr (D >= 3 && D <= DMAX && D > DMAX - 8) { \
        { if (layer0) check_node<(D >= 3 ? D : 3), true, false, false, TC>(lds_all, ent, jj, lb, mw, nm); else { if constexpr (!kPure) check_node<(D >= 3 ? D : 3), false, false, false, TC>(lds_all, ent, jj, lb, mw, nm); } } } break;
#define DVBS2_DEG_SWITCH switch (deg) { \
        DVBS2_DEG_CASE(3) DVBS2_DEG_CASE(4) DVBS2_DEG_CASE(5) DVBS2_DEG_CASE(6) DVBS2_DEG_CASE(7) DVBS2_DEG_CASE(8) \
        DVBS2_DEG_CASE(9) DVBS2_DEG_CASE(10) DVBS2_DEG_CASE(11) DVBS2_DEG_CASE(12) DVBS2_DEG_CASE(13) DVBS2_DEG_CASE(14) \
        DVBS2_DEG_CASE(15) DVBS2_DEG_CASE(16) DVBS2_DEG_CASE(17) DVBS2_DEG_CASE(18) DVBS2_DEG_CASE(19) DVBS2_DEG_CASE(20) \
        DVBS2_DEG_CASE(21) DVBS2_DEG_CASE(22) DVBS2_DEG_CASE(23) DVBS2_DEG_CASE(24) DVBS2_DEG_CASE(25) DVBS2_DEG_CASE(26) \
        DVBS2_DEG_CASE(27) DVBS2_DEG_CASE(28) DVBS2_DEG_CASE(29) DVBS2_DEG_CASE(30) DVBS2_DEG_CASE(31) DVBS2_DEG_CASE(32) \
        default: break; }

#define DVBS2_V2_CASE(D) case D: if constexpr (D >= 3 && D <= DMAX && D > DMAX - 8) { check_node_v2<(D >= 3 ? D : 3), DMAX, P6, TC>(ent, jj + lb, mw, nm, prefetch); } break;
#define DVBS2_V2_SWITCH switch (deg) { \
        DVBS2_V2_CASE(3) DVBS2_V2_CASE(4) DVBS2_V2_CASE(5) DVBS2_V2_CASE(6) DVBS2_V2_CASE(7) DVBS2_V2_CASE(8) \
        DVBS2_V2_CASE(9) DVBS2_V2_CASE(10) DVBS2_V2_CASE(11) DVBS2_V2_CASE(12) DVBS2_V2_CASE(13) DVBS2_V2_CASE(14) \
        DVBS2_V2_CASE(15) DVBS2_V2_CASE(16) DVBS2_V2_CASE(17) DVBS2_V2_CASE(18) DVBS2_V2_CASE(19) DVBS2_V2_CASE(20) \
        DVBS2_V2_CASE(21) DVBS2_V2_CASE(22) DVBS2_V2_CASE(23) DVBS2_V2_CASE(24) DVBS2_V2_CASE(25) DVBS2_V2_CASE(26) \
        DVBS2_V2_CASE(27) DVBS2_V2_CASE(28) DVBS2_V2_CASE(29) DVBS2_V2_CASE(30) DVBS2_V2_CASE(31) DVBS2_V2_CASE(32) \
        default: break; }

#define DVBS2_CHAIN_CASE(D) case D: if constexpr (D >= 4 && D <= DMAX && D > DMAX - 8) { check_node_chain_v2<(D >= 4 ? D : 4), DMAX, P6, TC>(ent, jj, jj + lb, work, block, mw, nm, htab16, hb_ctr, hb_epoch, hb_lane); } break;
#define DVBS2_CHAIN_SWITCH switch (deg) { \
        DVBS2_CHAIN_CASE(4) DVBS2_CHAIN_CASE(5) DVBS2_CHAIN_CASE(6) DVBS2_CHAIN_CASE(7) DVBS2_CHAIN_CASE(8) \
        DVBS2_CHAIN_CASE(9) DVBS2_CHAIN_CASE(10) DVBS2_CHAIN_CASE(11) DVBS2_CHAIN_CASE(12) DVBS2_CHAIN_CASE(13) DVBS2_CHAIN_CASE(14) \
        DVBS2_CHAIN_CASE(15) DVBS2_CHAIN_CASE(16) DVBS2_CHAIN_CASE(17) DVBS2_CHAIN_CASE(18) DVBS2_CHAIN_CASE(19) DVBS2_CHAIN_CASE(20) \
        DVBS2_CHAIN_CASE(21) DVBS2_CHAIN_CASE(22) DVBS2_CHAIN_CASE(23) DVBS2_CHAIN_CASE(24) DVBS2_CHAIN_CASE(25) DVBS2_CHAIN_CASE(26) \
        DVBS2_CHAIN_CASE(27) DVBS2_CHAIN_CASE(28) DVBS2_CHAIN_CASE(29) DVBS2_CHAIN_CASE(30) DVBS2_CHAIN_CASE(31) DVBS2_CHAIN_CASE(32) \
        default: break; }

// (A layer that runs the two-level lane chain gets an instantiation of its own -- TLC and the low-register form, which keeps one
// word per regular entry across the outer blocks instead of three --: compiled into the common instantiation the chain's register
// state made the compiler spill the regular entries of EVERY four- and eight-entry layer around it, 9/10 normal's multi-pair
// layers went from 12-17 k to 25-34 k cycles.)
#define DVBS2_HAZ_CALL1(D, NCV, LRV, TLCV) { \
        if (layer0) check_node_hazard<D, NCV, true, false, false, HZ2, LRV, TLCV, (MINW == 1), (DMAX <= 8), false, 0, false, TC>(lds_all, ent, jj, lb, work, block, block2, mw, nm, 0, nullptr, htab, hb_ctr, hb_epoch, hb_lane, hz_ph); else { if constexpr (!kPure) check_node_hazard<D, NCV, false, false, false, HZ2, LRV, TLCV, (MINW == 1), (DMAX <= 8), false, 0, false, TC>(lds_all, ent, jj, lb, work, block, block2, mw, nm, 0, nullptr, htab, hb_ctr, hb_epoch, hb_lane, hz_ph); } }
#define DVBS2_HAZ_CALL(D, NCV) { if constexpr (D - 2 >= NCV) { \
        if constexpr (kTlc<DMAX, HZ2> && !SOFT && MINW == 1 && (NCV == 4 || NCV == 8)) { if (block2 > 0 && htab != nullptr) DVBS2_HAZ_CALL1(D, NCV, (DMAX >= kTlcLowRegMinDmax), true) else DVBS2_HAZ_CALL1(D, NCV, false, false) } \
        else DVBS2_HAZ_CALL1(D, NCV, false, false) } }
#define DVBS2_HAZ_CASE(D) case D: if constexpr (D >= 4 && D <= DMAX && D > DMAX - 8) { \
        if (nc == 2) DVBS2_HAZ_CALL((D >= 4 ? D : 4), 2) else if (nc == 4) DVBS2_HAZ_CALL((D >= 4 ? D : 4), 4) else { if constexpr (HZ2 && DMAX <= kMaxHazard12Dmax) { if (nc == 8) DVBS2_HAZ_CALL((D >= 4 ? D : 4), 8) else DVBS2_HAZ_CALL((D >= 4 ? D : 4), 12) } else DVBS2_HAZ_CALL((D >= 4 ? D : 4), 8) } } break;
// The same with the packed first / last phase (check_node_hazard<..., V2P>): regular layers i > 0 of the builds with packed nodes whose
// wave record the host laid out in the packed format (header bit 14); the ordered phase is the plain one, instantiation for instantiation.
#define DVBS2_HAZP_CALL1(D, NCV, TLCV) { check_node_hazard<D, NCV, false, false, false, HZ2, false, TLCV, (MINW == 1), (DMAX <= 8), true, DMAX, P6, TC>(lds_all, ent, jj, lb, work, block, block2, mw, nm, 0, nullptr, htab, hb_ctr, hb_epoch, hb_lane, hz_ph); }
#define DVBS2_HAZP_CALL(D, NCV) { if constexpr (D - 2 >= NCV) { \
        if constexpr (kTlc<DMAX, HZ2> && !SOFT && MINW == 1 && (NCV == 4 || NCV == 8)) { if (block2 > 0 && htab != nullptr) DVBS2_HAZP_CALL1(D, NCV, true) else DVBS2_HAZP_CALL1(D, NCV, false) } \
        else DVBS2_HAZP_CALL1(D, NCV, false) } }
#define DVBS2_HAZP_CASE(D) case D: if constexpr (D >= 4 && D <= DMAX && D > DMAX - 8) { \
        if (nc == 2) DVBS2_HAZP_CALL((D >= 4 ? D : 4), 2) else if (nc == 4) DVBS2_HAZP_CALL((D >= 4 ? D : 4), 4) else DVBS2_HAZP_CALL((D >= 4 ? D : 4), 8) } break;
#define DVBS2_HAZP_SWITCH switch (deg) { \
        DVBS2_HAZP_CASE(4) DVBS2_HAZP_CASE(5) DVBS2_HAZP_CASE(6) DVBS2_HAZP_CASE(7) DVBS2_HAZP_CASE(8) \
        DVBS2_HAZP_CASE(9) DVBS2_HAZP_CASE(10) DVBS2_HAZP_CASE(11) DVBS2_HAZP_CASE(12) DVBS2_HAZP_CASE(13) DVBS2_HAZP_CASE(14) \
        DVBS2_HAZP_CASE(15) DVBS2_HAZP_CASE(16) DVBS2_HAZP_CASE(17) DVBS2_HAZP_CASE(18) DVBS2_HAZP_CASE(19) DVBS2_HAZP_CASE(20) \
        DVBS2_HAZP_CASE(21) DVBS2_HAZP_CASE(22) DVBS2_HAZP_CASE(23) DVBS2_HAZP_CASE(24) DVBS2_HAZP_CASE(25) DVBS2_HAZP_CASE(26) \
        DVBS2_HAZP_CASE(27) DVBS2_HAZP_CASE(28) DVBS2_HAZP_CASE(29) DVBS2_HAZP_CASE(30) DVBS2_HAZP_CASE(31) DVBS2_HAZP_CASE(32) \
        default: break; }
#define DVBS2_HAZ_SWITCH switch (deg) { \
        DVBS2_HAZ_CASE(4) DVBS2_HAZ_CASE(5) DVBS2_HAZ_CASE(6) DVBS2_HAZ_CASE(7) DVBS2_HAZ_CASE(8) \
        DVBS2_HAZ_CASE(9) DVBS2_HAZ_CASE(10) DVBS2_HAZ_CASE(11) DVBS2_HAZ_CASE(12) DVBS2_HAZ_CASE(13) DVBS2_HAZ_CASE(14) \
        DVBS2_HAZ_CASE(15) DVBS2_HAZ_CASE(16) DVBS2_HAZ_CASE(17) DVBS2_HAZ_CASE(18) DVBS2_HAZ_CASE(19) DVBS2_HAZ_CASE(20) \
        DVBS2_HAZ_CASE(21) DVBS2_HAZ_CASE(22) DVBS2_HAZ_CASE(23) DVBS2_HAZ_CASE(24) DVBS2_HAZ_CASE(25) DVBS2_HAZ_CASE(26) \
        DVBS2_HAZ_CASE(27) DVBS2_HAZ_CASE(28) DVBS2_HAZ_CASE(29) DVBS2_HAZ_CASE(30) DVBS2_HAZ_CASE(31) DVBS2_HAZ_CASE(32) \
        default: break; }

// MINW = 6 ("dense"): compiled for 80 VGPRs so that two pair-workgroups share a CU when the frames are short enough for
// LDS. It spills and only pays where ordered hazard steps dominate (ldpc_hip.hip picks it).
// V2: the packed nodes (check_node_v2, check_node_chain_v2) are compiled in; a table runs the build that measured faster for it
// (compiling both families into one kernel costs each of them 4-7 % through register allocation).
// SOLO: ONE frame per workgroup, two (or more) independent workgroups per CU. The pair workgroup exists only to put three
// waves on every SIMD; its price is that the hardware barrier couples the two frames, so each one also waits through the other's
// ordered hazard steps, LDS round trips and stragglers. Two separate 6-wave workgroups land 4,2,3,3 on the SIMDs
// (tools/ubench/placement.hip: waves go round the SIMDs, the next workgroup starts one position later). SOLO launches EIGHT waves
// -- always two per SIMD -- and lets two of them leave at once: a workgroup keeps both waves on one pair of SIMDs and one
// wave on the other pair, and workgroups sharing a CU take complementary patterns (a counter pair per CU in global memory).
// Result: three working waves per SIMD again, but the frames no longer wait for each other. Needs <= 128 VGPRs.
// Step 1 of the full syndrome test (see the kernel): the 360-bit sign vectors of all N / 360 groups, one thread per eight consecutive
// LLR bytes; returns non-zero when one of this thread's bytes is a zero LLR. A function of its own, NOT inlined: inlined, its loop
// perturbed the register allocation of the sweep and cost the never-converging batches -- where it never runs -- up to 4 % (S2X 154/180).
__device__ __attribute__((noinline)) int syndrome_sign_vectors(const lds_byte_t* lds, lds_u32_t* sv, int N, int tid, bool tc /*LLR bytes are two's complement*/)
{
    lds_byte_t* svb = reinterpret_cast<lds_byte_t*>(sv);
    int zero = 0;
#pragma unroll 2
    for (int blk = tid; blk < N / 8; blk += kHalf) {
        const v2u32 v = *reinterpret_cast<const lds_v2u_t*>(lds + 8 * blk);
        const uint32_t ob = tc ? 0u : 0x80808080u;
        const uint32_t xa = v.x ^ ob, xb = v.y ^ ob; // two's complement: zero bytes = zero LLRs
        zero |= (int)((((xa - 0x01010101u) & ~xa) | ((xb - 0x01010101u) & ~xb)) & 0x80808080u);
        // sign bit of byte i -> bit i (offset binary: negative <=> bit 7 clear): bits 0, 8, 16, 24 gathered by a multiply
        const uint32_t na = (((xa & 0x80808080u) >> 7) * 0x01020408u) >> 24, nb = (((xb & 0x80808080u) >> 7) * 0x01020408u) >> 24;
        const int g = blk / 45;
        svb[g * (kSvWords * 4) + (blk - 45 * g)] = (uint8_t)((na & 0xfu) | ((nb & 0xfu) << 4));
    }
    return zero;
}

constexpr int kSoloThreads = 512;
__device__ __forceinline__ uint32_t hw_cu_index()
{
    uint32_t hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    // HW_ID: simd_id[5:4] cu_id[11:8] sh_id[12] se_id[15:13]
    return ((((xcc & 0xfu) * 8u + ((hw >> 13) & 7u)) * 2u + ((hw >> 12) & 1u)) * 16u + ((hw >> 8) & 0xfu));
}
constexpr int kCuSlots = 16 * 8 * 2 * 16;

template <int DMAX, bool TIMING, int MINW = 1, bool V2 = false, bool SOLO = false, bool CHAIN = false /*V2 = false only: the packed chain node alone*/,
          bool HZ2 = false, bool SOFT = false /*SOFT: frame barriers in software -- a build of its own: the barrier state in every barrier of
                             every build cost the 80-VGPR build 25-30 % (54 -> 199 spilled VGPRs) and the degree classes 20..32 4-10 %*/
          /*HZ2: heavy hazard layers: twelve ordered entries, two-level walk (check_node_hazard); a build of its own because the
                             extra register state costs the degree classes 28 and 32 ten percent everywhere else (B11, S2X B21)*/>
__global__ __launch_bounds__(SOLO ? kSoloThreads : kThreads, SOLO ? 4 : MINW) void ldpc_layered_kernel(
    const uint32_t* __restrict__ recs, const uint32_t* __restrict__ wrecs /*per (layer, wave) sweep records*/,
    const int8_t* __restrict__ llr_in, uint8_t* __restrict__ state,
    uint32_t* __restrict__ msgs, int* __restrict__ iters, int* __restrict__ good, const int* __restrict__ target,
    int n_frames, int N, int K, int q, int cap, int stop_on_good /*bit 0: stop at a good syndrome, bit 1: software frame barriers, bit 2: group-synchronous stop*/,
    unsigned long long* __restrict__ tdbg, int* __restrict__ cu_slots,
    const DemapFused dm /*mode != 0: a fresh decode takes XFECFRAME symbols and demaps while loading (llr_in is null then)*/)
{
    unsigned long long tm_bar = 0, tm_body = 0, tm_conf = 0, tm_synd = 0, tm_sweep = 0, tm_load = 0, tm_s1 = 0;
#define TSTAMP(x) do { if (TIMING) { x = __builtin_readcyclecounter(); } } while (0)
    unsigned long long tA = 0, tB = 0, tC = 0, tS0 = 0, tS1 = 0;
    const bool fresh = llr_in != nullptr || dm.mode != 0;
    if (!fresh) { // resume launch: a workgroup whose frames are all at their target leaves before touching LDS
        const int fa = SOLO ? (int)blockIdx.x : 2 * (int)blockIdx.x, fb = SOLO ? fa : fa + 1;
        const bool ta = fa < n_frames && iters[fa] < target[fa];
        const bool tb = fb < n_frames && iters[fb] < target[fb];
        if (!ta && !tb) return;
    }
    TSTAMP(tA);
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_all[];
    constexpr int RS = rec_stride(DMAX);
    constexpr int RSW = V2 ? 6 * rec_stride_wave(DMAX) : rec_stride(DMAX); // dwords from one layer's sweep record to the next
    constexpr int MW = DMAX / 4; // message dwords per check (fixed per kernel variant)
    // "pure" packed builds (DVBS2_V2_PURE_MIN_DMAX; hardware barriers only): the plain nodes are compiled for layer 0 only -- the plain hazard
    // nodes of the degree class 32 cost the packed ones around them 5 % through register allocation --; the host runs such a build only
    // for tables whose every (layer > 0, wave) record fits the packed format (ldpc_hip.hip)
#ifndef DVBS2_V2_PURE_SOFT
#define DVBS2_V2_PURE_SOFT 1 // also the software-barrier packed build of the pure classes (S2X 154/180 131.8 -> 133.9 k with ten fix slots, round 5)
#endif
    constexpr bool kPure = V2 && (!SOFT || DVBS2_V2_PURE_SOFT) && v2_pure_class(DMAX);
    constexpr bool TC = V2 && DMAX <= DVBS2_TC_MAX_DMAX;        // LLR bytes in LDS as two's complement (see lds_rdx)
    constexpr uint32_t kObState = TC ? 0x80808080u : 0u;        // LDS bytes <-> the offset-binary state in HBM
    int solo_tid = (int)threadIdx.x;
    int solo_slot = -1, solo_pat = 0;
    if constexpr (SOLO) {
        // role election (the first words of LDS are scratch until the LLRs are loaded)
        volatile lds_i32_t* e = reinterpret_cast<volatile lds_i32_t*>((lds_byte_t*)lds_all); // [0..3] waves seen per SIMD, [4] workers so far, [5] pattern
        if (threadIdx.x < 8) e[threadIdx.x] = 0;
        __syncthreads();
        if (threadIdx.x == 0) {
            solo_slot = (int)hw_cu_index();
            int* w = cu_slots + solo_slot; // low half: workgroups with pattern 0 resident on this CU, high half: pattern 1
            int old = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), pat;
            do { pat = (old & 0xffff) <= (old >> 16) ? 0 : 1; }
            while (!__hip_atomic_compare_exchange_strong(w, &old, old + (pat ? 0x10000 : 1), __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            e[5] = pat; e[6] = solo_slot;
        }
        __syncthreads();
        solo_pat = e[5]; solo_slot = e[6];
        int widx = -1;
        if ((threadIdx.x & 63) == 0) {
            uint32_t hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            const int simd = (int)((hw >> 4) & 3u);
            const int rank = __hip_atomic_fetch_add(const_cast<lds_i32_t*>(e) + simd, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const bool two = ((simd >> 1) & 1) == solo_pat; // pattern 0 keeps both waves on SIMDs 0,1; pattern 1 on SIMDs 2,3
            if (two || rank == 0) widx = __hip_atomic_fetch_add(const_cast<lds_i32_t*>(e) + 4, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        widx = __builtin_amdgcn_readfirstlane(widx);
        __syncthreads();
        // a workgroup that was NOT placed two waves per SIMD elects fewer or more than six: the surplus leaves, a shortfall cannot
        // be repaired (it has not been observed; the launch would then miss rows) -- so insist on six by falling back to arrival order
        const int nworkers = e[4];
        if (nworkers != 6) widx = (int)(threadIdx.x >> 6) < 6 ? (int)(threadIdx.x >> 6) : -1;
        __syncthreads(); // everyone has read the election words; they may be overwritten now
        if (widx < 0 || widx >= 6) return; // the two spare waves leave (the hardware barrier no longer counts them)
        solo_tid = widx * 64 + (int)(threadIdx.x & 63);
    }
    const int half = SOLO ? 0 : __builtin_amdgcn_readfirstlane(threadIdx.x >= kHalf ? 1 : 0); // wave-uniform, and the compiler knows it
    const int tid = SOLO ? solo_tid : (int)threadIdx.x - half * kHalf;
    const int lb_rel = half * (int)half_lds_bytes(N);
    const int lb = lb_rel + lds_address_of(lds_all); // absolute LDS address of this frame's region
    lds_byte_t* lds = (lds_byte_t*)lds_all + lb_rel; // (typed pointers: see lds_byte_t)
    lds_u32_t* sv = reinterpret_cast<lds_u32_t*>(lds + N); // N % 8 == 0
    volatile lds_i32_t* flags = reinterpret_cast<volatile lds_i32_t*>(sv + sv_area_words(N)); // [0] bad-or, [1] finished, [2] pre-test failed, [3] full test needed
    volatile lds_i32_t* other_flags = reinterpret_cast<volatile lds_i32_t*>(
        (lds_byte_t*)lds_all + (1 - half) * (int)half_lds_bytes(N) + N + sv_area_words(N) * 4);
    const int f = SOLO ? (int)blockIdx.x : 2 * (int)blockIdx.x + half;
    const bool have_frame = f < n_frames;
    const int lane = tid & 63, wave = tid >> 6;
    const int wave_u = __builtin_amdgcn_readfirstlane(wave); // the same value, known to be uniform (scalar addressing of the records)
    const int NG = N / kM;
    const bool active = tid < kM;

    int it = 0, tgt = 0;
    bool finished = !have_frame; // this half has nothing (more) to do; it still takes part in every barrier
    if (have_frame) {
        tgt = target ? target[f] : cap;
        if (fresh && dm.mode != 0) {
            // demapper fused into the load: symbol s gives LLRs 2 s, 2 s + 1 (QPSK, natural order) or the three column positions
            // ra0 + s, ra1 + s, ra2 + s (8PSK de-interleaver), each stored at its place of the internal layout
            auto put = [&](int n, int8_t v) {
                int idx = n;
                if (n >= K) { const int r = n - K, jq = r / q; idx = K + kM * (r - jq * q) + jq; } // pty[360*i + j] = parity[q*j + i]
                lds[idx] = (uint8_t)v ^ (TC ? 0x00u : 0x80u);
            };
            const float N0 = dm.n0[dm.n0_count > 1 ? f : 0];
            const float2* src = reinterpret_cast<const float2*>(dm.syms) + (size_t)f * dm.n_syms;
            if (dm.mode == 1) {
                const float scalar = qpsk_scalar(N0);
                for (int sidx = tid; sidx < dm.n_syms; sidx += kHalf) {
                    const float2 v = src[sidx];
                    put(2 * sidx, qpsk_llr(v.x, scalar)); put(2 * sidx + 1, qpsk_llr(v.y, scalar));
                }
            } else {
                const float dp = psk8_dist_prec(N0);
                for (int sidx = tid; sidx < dm.n_syms; sidx += kHalf) {
                    const float2 v = src[sidx];
                    int8_t b0, b1, b2;
                    psk8_llr(v.x, v.y, dm.rr, dm.ri, dp, b0, b1, b2);
                    put(dm.ra0 + sidx, b0); put(dm.ra1 + sidx, b1); put(dm.ra2 + sidx, b2);
                }
            }
        } else if (fresh) {
            const uint2* src = reinterpret_cast<const uint2*>(llr_in + (size_t)f * N);
            for (int c = tid; c < N / 8; c += kHalf) {
                uint2 v = src[c];
                if constexpr (!TC) { v.x ^= 0x80808080u; v.y ^= 0x80808080u; }
                const int n = 8 * c;
                if (n < K) *reinterpret_cast<lds_v2u_t*>(lds + n) = (v2u32){ v.x, v.y }; // K % 8 == 0
                else {
                    // pty[360*i + j] = parity[q*j + i] (layered_decoder.hh:150-152)
                    int r = n - K;
                    int jq = r / q, iq = r - jq * q;
#pragma unroll
                    for (int b = 0; b < 8; b++) {
                        lds[K + kM * iq + jq] = (uint8_t)((b < 4 ? v.x >> (8 * b) : v.y >> (8 * (b - 4))) & 0xffu);
                        if (++iq == q) { iq = 0; ++jq; }
                    }
                }
            }
        } else {
            it = iters[f];
            if (it >= tgt) finished = true; // nothing to do for this frame in this pass
            else {
                const uint2* src = reinterpret_cast<const uint2*>(state + (size_t)f * N);
                for (int c = tid; c < N / 8; c += kHalf) { const uint2 v = src[c]; *reinterpret_cast<lds_v2u_t*>(lds + 8 * c) = (v2u32){ v.x ^ kObState, v.y ^ kObState }; } // (the state in HBM is offset binary in every build)
            }
        }
    }
    const bool untouched = finished; // never loaded: must not write state/iters/good back
    if (tid == 0) { flags[0] = 0; flags[2] = 0; flags[3] = 0; flags[1] = finished ? 1 : 0; flags[4] = 0; }
    // Frame barriers in software (bit 1 of the flag word; pair workgroups only): worth it for high-degree tables without hazard
    // layers -- few barriers, long layers: S2X B21 +12 %, S2X B10 +10 % -- and a loss where barriers are frequent (B4 -8 %: the
    // counter costs ~300 cycles per barrier against ~30 for s_barrier). Chosen per table by the host.
    constexpr bool soft_bar = SOFT; // (bit 1 of the flag word is what the host sets when it launches this build)
    volatile lds_i32_t* hb_ctr = soft_bar ? flags + 4 : nullptr; // frame barrier counter (frame_barrier)
    int hb_epoch = 0;
    const int hb_lane = lane;
    __syncthreads();
    TSTAMP(tB); tm_load = tB - tA;

    // Messages go through a buffer descriptor based at this frame's records: the per-lane offset (row * 4, plus the
    // word offset) is loop-invariant and the per-layer offset is a scalar operand, so a message access costs no
    // VALU address arithmetic (a flat 64-bit address costs two to four VALU instructions per access).
    uint32_t* msg_base = msgs + (size_t)(have_frame ? f : 0) * q * MW * kMsgStride;
    const __amdgpu_buffer_rsrc_t mrs = __builtin_amdgcn_make_buffer_rsrc(msg_base, 0, q * MW * kMsgStride * 4, 0x00020000);
    constexpr int kLayerBytes = MW * kMsgStride * 4;
#define MSG_LD(soff, w, r4) __builtin_amdgcn_raw_buffer_load_b32(mrs, (r4), (soff) + (w) * (kMsgStride * 4), 0)
#define MSG_ST(v, soff, w, r4) __builtin_amdgcn_raw_buffer_store_b32((v), mrs, (r4), (soff) + (w) * (kMsgStride * 4), 0)
    // six-bit message fields (msg_pair16): word w of a check of degree dg holds p6_fields(dg, w) fields; one or two fields are a
    // 16-bit access (row * 2 inside the word's 1536-byte slot), none is no access at all
#ifndef DVBS2_P6_DW
#define DVBS2_P6_DW 3 // fields from which a word is a dword access (experiments: 1 = never use 16-bit accesses)
#endif
    // measured (table B4, 4096 frames): 107 k frames/s with byte messages, 93 k with six-bit fields at the same traffic (dword accesses
    // only), 85-91 k with the traffic actually reduced by a quarter: the unpacking costs more than the bytes bring -- the regular layers
    // are limited by VALU issue and memory traffic at the same time. Kept behind this switch, off.
#ifndef DVBS2_P6
#define DVBS2_P6 0
#endif
    constexpr bool P6 = DVBS2_P6 != 0;
    auto msg_load = [&](uint32_t* dst, int soff, int r4, bool packed, int dg) {
#pragma unroll
        for (int w = 0; w < MW; w++) {
            if (P6 && packed) {
                const int nf = dg - 5 * w; // uniform
                if (nf >= DVBS2_P6_DW) dst[w] = MSG_LD(soff, w, r4);
                else if (nf >= 1) dst[w] = (uint32_t)(unsigned short)__builtin_amdgcn_raw_buffer_load_b16(mrs, r4 >> 1, soff + w * (kMsgStride * 4), 1 /* sc0: a sub-dword store does not update a line held in the vector L1 */);
                else dst[w] = 0u;
            }
            else dst[w] = MSG_LD(soff, w, r4);
        }
    };
    auto msg_store = [&](const uint32_t* src, int soff, int r4, bool packed, int dg) {
#pragma unroll
        for (int w = 0; w < MW; w++) {
            if (P6 && packed) {
                const int nf = dg - 5 * w;
                if (nf >= DVBS2_P6_DW) MSG_ST(src[w], soff, w, r4);
                else if (nf >= 1) __builtin_amdgcn_raw_buffer_store_b16((unsigned short)src[w], mrs, r4 >> 1, soff + w * (kMsgStride * 4), 0);
            }
            else MSG_ST(src[w], soff, w, r4);
        }
    };
    // bnl = 0 before the first update (layered_decoder.hh:27-31,149): a frame's first sweep (it == 0, in the first pass or
    // when a frame that stopped at once is resumed) takes offset-binary zero bytes instead of loading them -- no memset
    // of the record area, no read traffic in sweep 0
    bool is_good = false;

    for (;;) {
        TSTAMP(tS0);
        // ---- syndrome test (layered_decoder.hh:32-49, algorithms.hh:195-202): bad if any check has a zero
        // LLR or an odd number of negative LLRs. Barriers are taken by every thread; work only by halves that need it.
        const bool need_synd = !finished && ((stop_on_good & 1) || it >= tgt);
        // Pre-test (the reference's bad() also returns at the first failing check): the 360 checks of ONE layer,
        // tested edge by edge. A failure here is final; only a frame that passes pays for the full test below.
        // (Round 4 measured a "sticky" choice of the pre-tested layer -- the layer in which the last full test found an unsatisfied check
        // instead of `it mod q`, so that a nearly converged frame skips full tests: exact, and no measurable change at the operating point
        // of bench.py (282.8 k vs 282.9 k frames/s): the full test is ~3 % of an update in which it runs. Not kept.)
        if (need_synd && active) {
            const int i0 = it % q;
            const uint32_t* rec = recs + (size_t)i0 * RS;
            const int deg = (int)(rec[0] & 0xffu) + 2;
            uint32_t x = 0, z = 0;
            // per build, measured (round 6, interleaved A/B of whole tables against the edge-by-edge loop): 9/10 normal +4.7 %, 5/6 +4.0 %, 8/9 +2.0 %,
            // 4/5 +1.3 %, S2X 25/36 +1.6 %, 1/4 normal +1.0 %, short 3/5 (the 80-VGPR build) +45 %; 3/4 normal -2.5 % and short 2/3 -1.0 % (class 16),
            // S2X 154/180 -1.8 % (class 32 with software barriers), B4 and short 1/4 unchanged
            constexpr bool kPretestChunk = (DVBS2_PRETEST_CHUNK != 0) && DMAX != 16 && !(DMAX == 32 && SOFT);
            if constexpr (kPretestChunk) {
            // Round 6: FOUR edges per trip -- their eight record words in one scalar load, the four LDS reads in flight together. Edge by
            // edge the loop paid one scalar-cache round trip and one LDS round trip per edge (~300 cycles x 30 edges of a 9/10 normal
            // check: the syndrome phase was 19 k of its 342 k cycles per update, cycle stamps). (records are padded to DMAX entries,
            // DMAX is a multiple of four: the last trip never reads past its record; an edge past the degree reads byte `tid` and is ignored)
            for (int k0 = 0; k0 < deg; k0 += 4) {
                uint32_t e[8], v[4];
#pragma unroll
                for (int u = 0; u < 8; u++) e[u] = rec[4 + 2 * k0 + u];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const int a0 = tid + (int)e[2 * u] - ((uint32_t)tid < e[2 * u + 1] ? 0 : kM);
                    v[u] = (uint32_t)lds[k0 + u < deg ? a0 : tid];
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    uint32_t w = v[u] ^ (TC ? 0x80u : 0x00u); // offset-binary value
                    if (i0 == 0 && k0 + u == deg - 1 && tid == 0) w = 0x81u; // check (0,0) has no previous parity: neutral +1
                    if (k0 + u < deg) { x ^= w; z |= (w == 0x80u); }
                }
            }
            } else
            for (int k = 0; k < deg; k++) {
                const int a0 = tid + (int)rec[4 + 2 * k] - ((uint32_t)tid < rec[5 + 2 * k] ? 0 : kM);
                uint32_t v = (uint32_t)lds[a0] ^ (TC ? 0x80u : 0x00u); // offset-binary value
                if (i0 == 0 && k == deg - 1 && tid == 0) v = 0x81u; // check (0,0) has no previous parity: neutral +1
                x ^= v;
                z |= (v == 0x80u);
            }
            // offset binary: the sign bit is inverted, so the count of negatives is deg - popcount(bit 7)
            const int bad_pre = (int)((((x >> 7) ^ (uint32_t)deg) & 1u) | z);
            if (__ballot(bad_pre) != 0 && lane == 0) flags[2] = 1;
        }
        lds_barrier();
#ifdef DVBS2_EXP_ALWAYS_FULL // timing experiment (same results): the full test after every update, whatever the pre-test says
        const bool need_full = need_synd;
#else
        const bool need_full = need_synd && flags[2] == 0;
#endif
        if (tid == 0) flags[3] = need_full ? 1 : 0;
        lds_barrier();
        const bool full_any = flags[3] != 0 || (!soft_bar && !SOLO && other_flags[3] != 0); // uniform over the barrier domain
        if (full_any) {
        if (need_full) {
            // Step 1: 360-bit sign vector per group. Round 3: one thread per EIGHT consecutive LLR bytes (one 8-byte LDS read -> one
            // byte of the vector: 360 = 45 x 8, so a block never straddles two groups) instead of one byte read and two ballots per
            // thread and group -- a fifth of the LDS instructions; at the operating point, where nearly every test is a full one,
            // the tests were a tenth of the decode (cycle stamps).
            {
                const int zero = syndrome_sign_vectors(lds, sv, N, tid, TC);
                if (__ballot(zero != 0) != 0 && lane == 0) flags[0] = 1;
            }
        }
        TSTAMP(tC); tm_s1 += tC - tS0;
        lds_barrier();
        if (need_full && tid < NG) { // wrap extension: bits 360+u = bit u
            lds_u32_t* p = sv + tid * kSvWords;
            const uint32_t w0 = p[0], w1 = p[1];
            p[11] = (p[11] & 0xffu) | (w0 << 8);
            p[12] = (w0 >> 24) | (w1 << 8);
        }
        lds_barrier();
        if (need_full) {
            // Step 2: parity word (layer i, lanes 32w..32w+31) = xor over entries of the rotated sign vectors
            // (this one stays inline: as a function of its own its record arrays made the degree classes 12-dense and 32 two to
            // three times slower -- the kernel then has to provide the callee's registers on top of its own)
            int bad = 0;
            for (int item = tid; item < q * 12; item += kHalf) {
                const int i = item / 12, w = item - 12 * i;
                const uint32_t* rec = recs + (size_t)i * RS;
                const int deg = (int)(rec[0] & 0xffu) + 2;
                uint32_t e0[DMAX], e1[DMAX];
#pragma unroll
                for (int k = 0; k < DMAX; k++) { e0[k] = rec[4 + 2 * k]; e1[k] = rec[5 + 2 * k]; } // padded records: always readable
                uint32_t acc = 0;
#pragma unroll
                for (int k = 0; k < DMAX; k++) {
                    if (k < deg) {
                        const int rot = kM - (int)e1[k];   // S0 = 360*g + rot, thr = 360 - rot
                        const int g360 = (int)e0[k] - rot;
                        const int t0 = wrap360(32 * w + rot);
                        const lds_u32_t* p = sv + (g360 / kM) * kSvWords + (t0 >> 5);
                        uint32_t x = __funnelshift_r(p[0], p[1], t0 & 31);
                        if (i == 0 && k == deg - 1 && w == 0) x &= ~1u; // check (0,0): no previous parity
                        acc ^= x;
                    }
                }
                if (w == 11) acc &= 0xffu;
                bad |= acc != 0;
            }
            if (__ballot(bad) != 0 && lane == 0) flags[0] = 1;
        }
        lds_barrier();
        } // full_any
        if (need_synd) is_good = need_full && flags[0] == 0;
        const bool gs = (stop_on_good & 5) == 5; // group-synchronous stop (group_decide), first pass only
        if (!finished && (it >= tgt || (!gs && (stop_on_good & 1) && is_good))) finished = true;
        lds_barrier(); // everyone has read flags[0]
        if (wave_u == 0) { // (the whole first wave: group_decide reads one status word per lane)
            int fin = finished ? 1 : 0;
            if (gs && !finished) {
                const uint32_t* hd = recs - kRecHeaderWords; // (uniform: scalar loads)
                const int* iters0 = reinterpret_cast<const int*>(((unsigned long long)hd[1] << 32) | hd[0]);
                int* status = reinterpret_cast<int*>(((unsigned long long)hd[3] << 32) | hd[2]);
                const int G = (int)hd[4];
                const int g = f / G; // within this launch (its first frame is a multiple of the group size)
                fin = group_decide(status + (iters - iters0) + g * G, min(G, n_frames - g * G), f - g * G, it, is_good, lane, (int)hd[5]) != 0;
            }
            if (tid == 0) { flags[0] = 0; flags[2] = 0; flags[1] = fin; }
        }
        lds_only_barrier();
        if (gs) finished = flags[1] != 0; // (uniform over the frame)
        TSTAMP(tS1); tm_synd += tS1 - tS0;
        if (finished && (soft_bar || SOLO || other_flags[1])) break; // uniform over the barrier domain

        // ---- one update sweep: layered_decoder.hh:50-79 ----
        // Threads 360..383 of a half mirror check row 359: same reads, same results, same (duplicate) writes. That
        // keeps the whole sweep free of per-lane predicates: `work` is wave-uniform.
        const bool work = __builtin_amdgcn_readfirstlane(finished ? 0 : 1) != 0; // uniform over the wave, and the compiler knows it
        const int row = tid < kM ? tid : kM - 1;
        const int row4 = row * 4;
        const bool zero_msgs = it == 0; // uniform over the half
        // timing builds: cycles per phase of the hazard nodes, frame 0, lane 0 of waves 0 and 5 (slots 256.. and 272.. after the per-layer sums)
        unsigned long long* const hz_ph = (TIMING && tdbg && f == 0 && (tid == 0 || tid == 320)) ? tdbg + (size_t)n_frames * 48 + 256 + (tid ? 16 : 0) : nullptr;
        uint32_t pre[MW]; // messages of the next layer for check tid, loaded one layer ahead
        constexpr bool kWaitStore = (DVBS2_WAIT_BEFORE_STORE != 0) && !SOFT;
        if (work) {
#pragma unroll
            for (int w = 0; w < MW; w++) {
                pre[w] = zero_msgs ? 0x80808080u : MSG_LD(0, w, row4);
            }
        }
        // Layer records are double-buffered in SGPRs: the scalar loads of layer i+1 are issued at the top of layer i
        // (an un-prefetched s_load at the head of every layer was a quarter of the sweep time). Small records are
        // buffered whole; for the large ones only every 8th dword is carried over -- enough to pull each cache
        // line of the next record into the scalar cache -- and the rest is loaded at the top of the layer.
        // (Rounds 1-3 carried every 8th word of the large records over from the previous layer; round 4 measured none -- the header words only --
        // 1-9 % faster on 16 of 17 tables of the classes 16-32: each carried word was its own `s_load_dword; s_waitcnt lgkmcnt(0); v_writelane`
        // at every layer head, and the records of a table (5-7 KB) stay in the scalar cache anyway.)
        constexpr int PF = DMAX <= DVBS2_PF_SMALL_MAX_DMAX ? 1 : 2 * DMAX;
        // the sweep reads the records of its own WAVE (check_node_v2): wrecs[(layer * 6 + wave) * RS]
        const uint32_t* wr = V2 ? wrecs + (size_t)wave_u * rec_stride_wave(DMAX) : recs; // builds without packed nodes read the per-layer records
        uint32_t nhdr = wr[0], ninfo = wr[1]; // word 1: message format of the NEXT layer for this wave (degree | packed << 8)
        uint32_t nent[2 * DMAX];
#pragma unroll
        for (int k = 0; k < 2 * DMAX; k += PF) nent[k] = wr[4 + k]; // (PF = 2 DMAX: word 0 only, never used)
        DVBS2_WAIT_VM0(); // (the first layer's messages: once per sweep, so that inside the loop no path has a load pending at a layer boundary)
#if DVBS2_WAIT_RECORDS
        // the same for the first layer's RECORD: with its scalar loads pending on the path into the loop the compiler waits for them at the
        // first use of the header -- behind the next record's prefetch, which every iteration then waits for as soon as it has issued it
        // (an empty asm statement that "uses" the loaded words: the compiler has to complete the loads in front of it; an explicit s_waitcnt
        // alone does not hold them -- loads of constant memory are moved across it)
        asm volatile("" : "+s"(nhdr), "+s"(ninfo), "+s"(nent[0]));
#endif
        for (int i = 0; i < q; i++) {
            const uint32_t hdr = nhdr, info = ninfo;
            const bool npacked = (info >> 8) & 1u; const int ndeg = (int)(info & 0xffu);
            uint32_t ent[2 * DMAX];
#pragma unroll
            for (int k = 0; k < 2 * DMAX; k++) ent[k] = (PF <= 2 * DMAX - 1 && k % PF == 0) ? nent[k] : wr[(size_t)i * RSW + 4 + k];
            const uint32_t* nrec = wr + (size_t)(i + 1 < q ? i + 1 : 0) * RSW;
            auto prefetch = [&](uint32_t after) {
                const uint32_t* p = nrec; (void)after;
                nhdr = p[0]; ninfo = p[1];
                if constexpr (PF <= 2 * DMAX - 1) {
#pragma unroll
                    for (int k = 0; k < 2 * DMAX; k += PF) nent[k] = p[4 + k];
                }
            };
            // a packed-node layer (bit 13) may issue these loads from inside the node; the others here
            // Degree classes up to DVBS2_PREFETCH_AFTER_BARRIER_MAX_DMAX issue them BEHIND the layer's barrier: in front of it the wave waits
            // for them (lgkmcnt(0) of the barrier and of the first use of the header) as soon as it has issued them. Measured: B4 +0.7 %,
            // 1/3 normal +1.5 %, S2X 9/20 +1.2 %, 1/4 normal +1.4 %; the classes 12-32 lose up to 5 % (S2X_TABLE_B16) -- class 8 only.
            constexpr bool kPab = DMAX <= DVBS2_PREFETCH_AFTER_BARRIER_MAX_DMAX;
            if constexpr (!kPab) prefetch(0u);
            const int deg = (int)(hdr & 0xffu) + 2;
            const int nc = (int)((hdr >> 8) & 0xfu);
            lds_u32_t* htab = ((hdr >> 12) & 1u) ? sv : nullptr; // lane-chain scratch: the sign-vector area is idle during a sweep
            const int block = (int)(hdr >> 16);
            int block2 = 0; // hazard layers: rows per outer block of the two-level walk (0: off)
            if constexpr (HZ2 || (kTlc<DMAX, HZ2> && !SOFT && MINW == 1)) { if (block < kM) block2 = (int)wr[(size_t)i * RSW + 2]; }
            const bool layer0 = (i == 0);
            const int mso = i * kLayerBytes; // scalar byte offset of this layer's message records
            TSTAMP(tA);
            if (hdr & 0x8000u) lds_barrier();
            if constexpr (kPab) { asm volatile("" ::: "memory"); prefetch(0u); }
            TSTAMP(tB); tm_bar += tB - tA;
            if (block >= kM) {
                // regular layer: all 360 checks at once
                if (work) {
                    const int jj = row;
                    const bool v2 = V2 && ((hdr >> 13) & 1u);
                    // v2: this wave's record is in the packed node's format (two's complement messages)
                    uint32_t mw[MW], nm[MW];
#pragma unroll
                    for (int w = 0; w < MW; w++) mw[w] = zero_msgs ? (v2 ? 0u : 0x80808080u) : pre[w];
                    if (i + 1 < q && !zero_msgs) msg_load(pre, mso + kLayerBytes, row4, npacked, ndeg);
                    if constexpr (V2) { if (v2) { DVBS2_V2_SWITCH } else DVBS2_DEG_SWITCH } else DVBS2_DEG_SWITCH
                    DVBS2_WAIT_VM0();
                    msg_store(nm, mso, row4, v2, deg);
                }
                TSTAMP(tC); tm_body += tC - tB;
                if (TIMING && tdbg && f == 0 && tid == 0) tdbg[(size_t)n_frames * 48 + i] += tC - tA; // per-layer cycles of frame 0, wave 0 (incl. its barrier)
            } else {
                if (nc != kHazardWalk) {
                    // sequential-order hazard inside the layer: check_node_hazard (every thread takes every barrier)
                    const int jj = row;
                    // hv2: packed single-pair chain (check_node_chain_v2): two's complement messages. In a build without the packed regular
                    // node (CHAIN) the layer's per-wave record is fetched here, from wrecs, when the per-layer record says so (bit 14).
                    const bool hv2 = (V2 && ((hdr >> 13) & 1u)) || (!V2 && CHAIN && ((hdr >> 14) & 1u));
                    if constexpr (!V2 && CHAIN) {
                        if (hv2) {
                            const uint32_t* cw = wrecs + ((size_t)i * 6 + wave_u) * rec_stride_wave(DMAX) + 4;
#pragma unroll
                            for (int k = 0; k < 2 * DMAX; k++) ent[k] = cw[k];
                        }
                    }
                    uint32_t mw[MW], nm[MW];
#pragma unroll
                    for (int w = 0; w < MW; w++) mw[w] = (work && !zero_msgs) ? pre[w] : (hv2 ? 0u : 0x80808080u);
                    if (work && i + 1 < q && !zero_msgs) msg_load(pre, mso + kLayerBytes, row4, npacked, ndeg);
                    // hvp: the generic hazard node with the packed first / last phase (header bit 14 of this wave's record; every wave of the
                    // layer runs the same ordered phase, whichever form its own record has)
                    const bool hvp = V2 && v2p_class(DMAX) && ((hdr >> 14) & 1u);
                    if constexpr ((V2 || CHAIN) && DMAX <= 16) { // (the chain node's register state costs the high-degree builds more than it saves: not built there)
                        if (hv2 && !hvp) {
                            lds_u32_t* htab16 = lds_align16<lds_u32_t>(sv); // 16-byte records
                            DVBS2_CHAIN_SWITCH
                        } else if constexpr (V2 && v2p_class(DMAX)) { if (hvp) { DVBS2_HAZP_SWITCH } else DVBS2_HAZ_SWITCH }
                        else DVBS2_HAZ_SWITCH
                    } else if constexpr (V2 && v2p_class(DMAX)) { if (hvp) { DVBS2_HAZP_SWITCH } else DVBS2_HAZ_SWITCH }
                    else DVBS2_HAZ_SWITCH
                    DVBS2_WAIT_VM0(); // (on every path, so that nothing is pending behind it whatever the branch)
                    if (work) msg_store(nm, mso, row4, hv2, deg);
                } else {
                    // too many hazard entries: the first wave of the half walks the 360 checks alone in ascending
                    // chunks of min(B_i, 64) (LDS operations of one wave execute in order: no barrier between chunks)
                    if (!finished && wave == 0) {
                        const int chunk = block < 64 ? block : 64;
                        for (int start = 0; start < kM; start += chunk) {
                            const int jj = start + lane;
                            if (lane < chunk && jj < kM) {
                                uint32_t mw[MW], nm[MW];
#pragma unroll
                                for (int w = 0; w < MW; w++) mw[w] = zero_msgs ? 0x80808080u : MSG_LD(mso, w, jj * 4);
                                DVBS2_DEG_SWITCH
#pragma unroll
                                for (int w = 0; w < MW; w++) MSG_ST(nm[w], mso, w, jj * 4);
                            }
                        }
                    }
                    lds_barrier();
                    if (work && i + 1 < q && !zero_msgs) msg_load(pre, mso + kLayerBytes, row4, npacked, ndeg);
                    DVBS2_WAIT_VM0(); // (rare path; keeps "nothing pending at the end of a layer" true on EVERY path, see DVBS2_WAIT_BEFORE_STORE)
                }
                TSTAMP(tC); tm_conf += tC - tB;
                if (TIMING && tdbg && f == 0 && tid == 0) tdbg[(size_t)n_frames * 48 + i] += tC - tA;
            }
        }
        TSTAMP(tA);
        lds_barrier();
        TSTAMP(tB); tm_bar += tB - tA; tm_sweep += tB - tS1;
        if (!finished) it++;
    }
    if (TIMING && tdbg && lane == 0 && have_frame) {
        unsigned long long* o = tdbg + ((size_t)f * 6 + wave) * 8;
        o[0] = tm_load; o[1] = tm_synd; o[2] = tm_sweep; o[3] = tm_bar; o[4] = tm_body; o[5] = tm_conf; o[6] = (unsigned long long)it; o[7] = tm_s1;
    }

    if (have_frame && !untouched) {
        if (tid == 0) { iters[f] = it; good[f] = is_good ? 1 : 0; }
        uint2* dst = reinterpret_cast<uint2*>(state + (size_t)f * N);
        for (int c = tid; c < N / 8; c += kHalf) { const v2u32 v = *reinterpret_cast<const lds_v2u_t*>(lds + 8 * c); dst[c] = make_uint2(v.x ^ kObState, v.y ^ kObState); }
    }
    if constexpr (SOLO) { // give the CU's pattern slot back
        if (tid == 0) __hip_atomic_fetch_add(cu_slots + solo_slot, solo_pat ? -0x10000 : -1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}


// ---- host-side launch interface of one kernel variant (defined in ldpc_inst_*.hip) ----
struct LdpcLaunch {
    const uint32_t* recs; const uint32_t* wrecs; const int8_t* llr_in; uint8_t* state; uint32_t* msgs; int* iters; int* good; const int* target;
    int n_frames, N, K, q, cap, stop_on_good; unsigned long long* tdbg;
    DemapFused dm;
    size_t lds_bytes; hipStream_t stream;
    bool dense; // the 80-VGPR build of the kernel (two workgroups per CU), see kDenseBuilt
    bool v2;    // the build with the packed nodes
    bool solo;  // one frame per workgroup (kSoloBuilt)
    bool chain; // plain build + packed chain node (kChainBuilt; ignored with v2, which has it anyway)
    bool hz2;   // plain pair build with the heavy-hazard paths (kHz2Built)
    bool soft;  // pair build (plain or packed) with software frame barriers (kSoftBuilt)
    bool pr_packed = false; // parity-in-records kernel (ldpc_kernel_pr.hpp): packed nodes in the regular middle layers (two-dword records; v2 there = one-dword records)
    int* cu_slots;
};
template <int DMAX> hipError_t ldpc_variant_prepare(size_t pair_lds_bytes, size_t solo_lds_bytes);
template <int DMAX> void ldpc_variant_launch(const LdpcLaunch& a);
#ifndef DVBS2_SOLO_MAX_DMAX
#define DVBS2_SOLO_MAX_DMAX 16 // (experiments: one-frame workgroups -- 128 VGPRs -- for higher degree classes; round 6 bound, notes/r06_experiments.md)
#endif
constexpr int kSoloMaxDmax = DVBS2_SOLO_MAX_DMAX;
template <int DMAX> constexpr bool kSoloBuilt = (DMAX <= kSoloMaxDmax);
// plain builds with the packed chain node: measured SLOWER than the plain build's own lane chain (B4 107.8 k vs 109.8 k, B5 57.9 k vs
// 62.2 k frames/s) although its ordered steps cost a third -- the node's register state hurts the rest of the kernel. Not built.
template <int DMAX> constexpr bool kHz2Built = (DMAX >= 12);
template <int DMAX> constexpr bool kSoftBuilt = (DMAX >= 20); // pays where layers are long and barriers few (measured: S2X B10, B20, B21, B24)
template <int DMAX> constexpr bool kChainBuilt = false; // 128 VGPRs: four waves per SIMD must fit while a workgroup starts

#ifdef DVBS2_LDPC_INSTANTIATE
// The cycle-stamped variant (DVBS2_TIMING=1, tools/exp_tables.py) is only built for DMAX = 8 -- the headline tables --
// to keep the build time of the large variants down; elsewhere the request is ignored.
#ifdef DVBS2_TIMING_ALL
template <int DMAX> constexpr bool kTimingBuilt = true; // experiment builds (tools/build_variant.sh timing -DDVBS2_TIMING_ALL)
#else
template <int DMAX> constexpr bool kTimingBuilt = (DMAX == 8);
#endif
// only the degree class 5..12 survives 80 VGPRs (120 B of scratch); the classes of short 5/6 and 8/9 (DMAX 20, 28) spill so
// much that they run 8x slower (measured)
template <int DMAX> constexpr bool kDenseBuilt = (DMAX == 12);
#ifdef DVBS2_TIMING_HZ2 // experiment builds: cycle stamps in the heavy-hazard build instead of the packed one
#define DVBS2_TIMING_KERNEL ldpc_layered_kernel<DMAX, true, 1, false, false, false, true>
#else
#define DVBS2_TIMING_KERNEL ldpc_layered_kernel<DMAX, true, 1, true, false>
#endif
#define DVBS2_KARGS a.recs, a.wrecs, a.llr_in, a.state, a.msgs, a.iters, a.good, a.target, a.n_frames, a.N, a.K, a.q, a.cap, a.stop_on_good
template <int DMAX> hipError_t ldpc_variant_prepare(size_t pair_lds_bytes, size_t solo_lds_bytes)
{
    auto set = [](const void* k, size_t b) { return hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b); };
    hipError_t e = set((const void*)ldpc_layered_kernel<DMAX, false, 1, false, false>, pair_lds_bytes);
    if (e == hipSuccess) e = set((const void*)ldpc_layered_kernel<DMAX, false, 1, true, false>, pair_lds_bytes);
    if constexpr (kSoloBuilt<DMAX>) {
        if (e == hipSuccess) e = set((const void*)ldpc_layered_kernel<DMAX, false, 1, false, true>, solo_lds_bytes);
        if (e == hipSuccess) e = set((const void*)ldpc_layered_kernel<DMAX, false, 1, true, true>, solo_lds_bytes);
    }
    if constexpr (kChainBuilt<DMAX>) {
        if (e == hipSuccess) e = set((const void*)ldpc_layered_kernel<DMAX, false, 1, false, false, true>, pair_lds_bytes);
        if (e == hipSuccess) e = set((const void*)ldpc_layered_kernel<DMAX, false, 1, false, true, true>, solo_lds_bytes);
    }
    if constexpr (kHz2Built<DMAX>) if (e == hipSuccess) e = set((const void*)ldpc_layered_kernel<DMAX, false, 1, false, false, false, true>, pair_lds_bytes);
    if constexpr (kSoftBuilt<DMAX>) {
        if (e == hipSuccess) e = set((const void*)ldpc_layered_kernel<DMAX, false, 1, false, false, false, false, true>, pair_lds_bytes);
        if (e == hipSuccess) e = set((const void*)ldpc_layered_kernel<DMAX, false, 1, true, false, false, false, true>, pair_lds_bytes);
    }
    if constexpr (kDenseBuilt<DMAX>) if (e == hipSuccess) e = set((const void*)ldpc_layered_kernel<DMAX, false, 6, false, false>, pair_lds_bytes);
    if constexpr (kTimingBuilt<DMAX>) if (e == hipSuccess) e = set((const void*)DVBS2_TIMING_KERNEL, pair_lds_bytes);
    return e;
}
template <int DMAX> void ldpc_variant_launch(const LdpcLaunch& a)
{
    const dim3 grid((a.n_frames + 1) / 2), block(kThreads);
    if constexpr (kTimingBuilt<DMAX>) {
        if (a.tdbg) {
            hipLaunchKernelGGL((DVBS2_TIMING_KERNEL), grid, block, a.lds_bytes, a.stream, DVBS2_KARGS, a.tdbg, nullptr, a.dm);
            return;
        }
    }
    if constexpr (kDenseBuilt<DMAX>) {
        if (a.dense) {
            hipLaunchKernelGGL((ldpc_layered_kernel<DMAX, false, 6, false, false>), grid, block, a.lds_bytes, a.stream, DVBS2_KARGS, nullptr, nullptr, a.dm);
            return;
        }
    }
    if constexpr (kHz2Built<DMAX>) {
        if (a.hz2) {
            hipLaunchKernelGGL((ldpc_layered_kernel<DMAX, false, 1, false, false, false, true>), grid, block, a.lds_bytes, a.stream, DVBS2_KARGS, nullptr, nullptr, a.dm);
            return;
        }
    }
    if constexpr (kSoftBuilt<DMAX>) {
        if (a.soft && !a.solo) {
            if (a.v2) hipLaunchKernelGGL((ldpc_layered_kernel<DMAX, false, 1, true, false, false, false, true>), grid, block, a.lds_bytes, a.stream, DVBS2_KARGS, nullptr, nullptr, a.dm);
            else hipLaunchKernelGGL((ldpc_layered_kernel<DMAX, false, 1, false, false, false, false, true>), grid, block, a.lds_bytes, a.stream, DVBS2_KARGS, nullptr, nullptr, a.dm);
            return;
        }
    }
    if constexpr (kSoloBuilt<DMAX>) {
        if (a.solo) {
            const dim3 sgrid(a.n_frames), sblock(kSoloThreads);
            if (a.v2) hipLaunchKernelGGL((ldpc_layered_kernel<DMAX, false, 1, true, true>), sgrid, sblock, a.lds_bytes, a.stream, DVBS2_KARGS, nullptr, a.cu_slots, a.dm);
            else if (a.chain) hipLaunchKernelGGL((ldpc_layered_kernel<DMAX, false, 1, false, true, true>), sgrid, sblock, a.lds_bytes, a.stream, DVBS2_KARGS, nullptr, a.cu_slots, a.dm);
            else hipLaunchKernelGGL((ldpc_layered_kernel<DMAX, false, 1, false, true>), sgrid, sblock, a.lds_bytes, a.stream, DVBS2_KARGS, nullptr, a.cu_slots, a.dm);
            return;
        }
    }
    if (a.v2) hipLaunchKernelGGL((ldpc_layered_kernel<DMAX, false, 1, true, false>), grid, block, a.lds_bytes, a.stream, DVBS2_KARGS, nullptr, nullptr, a.dm);
    else if constexpr (kChainBuilt<DMAX>) {
        if (a.chain) hipLaunchKernelGGL((ldpc_layered_kernel<DMAX, false, 1, false, false, true>), grid, block, a.lds_bytes, a.stream, DVBS2_KARGS, nullptr, nullptr, a.dm);
        else hipLaunchKernelGGL((ldpc_layered_kernel<DMAX, false, 1, false, false>), grid, block, a.lds_bytes, a.stream, DVBS2_KARGS, nullptr, nullptr, a.dm);
    }
    else hipLaunchKernelGGL((ldpc_layered_kernel<DMAX, false, 1, false, false>), grid, block, a.lds_bytes, a.stream, DVBS2_KARGS, nullptr, nullptr, a.dm);
}
template hipError_t ldpc_variant_prepare<DVBS2_LDPC_INSTANTIATE>(size_t, size_t);
template void ldpc_variant_launch<DVBS2_LDPC_INSTANTIATE>(const LdpcLaunch&);
#endif

} // namespace dvbs2
