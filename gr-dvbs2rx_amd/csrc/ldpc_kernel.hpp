// ldpc_kernel.hpp -- device code of the layered LDPC decoder (included by ldpc_hip.hip for the constants and by
// the per-variant translation units ldpc_inst_*.hip, which instantiate one DMAX each so that the seven kernel
// variants compile in parallel).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

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
__host__ __device__ constexpr int rec_stride(int dmax) { return 2 * dmax + 4; }
__host__ __device__ constexpr size_t half_lds_bytes(int N) { return ((size_t)N + (size_t)(N / kM) * kSvWords * 4 + 32 + 15) / 16 * 16; }

__device__ __forceinline__ int wrap360(int t) { return t >= kM ? t - kM : t; }

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
typedef __attribute__((address_space(3))) uint8_t lds_byte_t;
__device__ __forceinline__ int lds_rd(int a) { return *reinterpret_cast<const lds_byte_t*>((size_t)(uint32_t)a); }
__device__ __forceinline__ void lds_wr(int a, int v) { *reinterpret_cast<lds_byte_t*>((size_t)(uint32_t)a) = (uint8_t)v; }
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
template <int DEG, bool LAYER0, bool PR = false, bool LAST = false>
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
        else Lb[k] = lds_rd(ad[k]);
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
        else if (!(LAYER0 && k == DEG - 1) || last_valid) lds_wr(ad[k], nl);
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
constexpr int kLaneChainMaxDeg = 28;                                  // not instantiated for the big variants nor for the
                                                                      // 80-VGPR parity-in-records kernel (registers)
constexpr int kMaxHazard = 8;
constexpr int kHazardWalk = 15; // header code: too many hazard entries, fall back to the single-wave chunk walk
template <int DEG, int NC, bool LAYER0, bool PR = false, bool LAST = false>
__device__ __forceinline__ void check_node_hazard(uint8_t* __restrict__ lds, const uint32_t* ent, int jj, int lb, bool work,
                                                  int block, const uint32_t* mw, uint32_t* nm, int own_in = 0, int* carry = nullptr,
                                                  uint32_t* tab = nullptr /*lane_chain_words(block) of LDS scratch when the layer is a lane chain*/)
{
    constexpr bool OWN_REG = PR && !LAST;     // entry DEG-2 (see check_node)
    constexpr bool PREV_REG = PR && !LAYER0;  // entry DEG-1
    int ad[DEG], inp[DEG], mg[DEG];
    const int jjb = jj + lb, jjb360 = jjb - kM;
    int min0 = 127, min1 = 127, signs = 0;
    int spare = 0x80;
    const bool last_valid = !LAYER0 || jj != 0;
    __builtin_amdgcn_s_setprio(0); // as in check_node; the ordered steps below run at the top priority
    if (work) {
#pragma unroll
        for (int k = 0; k < DEG; k++) {
            if (k >= DEG - 2 && !(LAYER0 && k == DEG - 1)) ad[k] = jjb + (int)ent[2 * k];
            else ad[k] = wrap_addr(jj, jjb, jjb360, ent[2 * k], ent[2 * k + 1]);
        }
#pragma unroll
        for (int k = 0; k < DEG; k++) {
            if (k >= NC) { // regular entry
                const int Lb = (OWN_REG && k == DEG - 2) ? own_in : (PREV_REG && k == DEG - 1) ? *carry : lds_rd(ad[k]);
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
#pragma unroll
    for (int w = 0; w < (DEG + 3) / 4; w++) nm[w] = 0;
    // P2 keeps only what the NEXT block needs on its critical path: the new hazard LLRs. For hazard entry k the
    // magnitude sent back is the minimum over all OTHER entries = min(partial min0 of the regular entries, the
    // other hazard magnitudes) and the sign is the xor of all other signs; the merge of the hazard entries into
    // (min0, min1, signs) for P3 and the hazard message bytes are computed after the loop.
    if (PR && (DEG + 3) / 4 < 2) nm[1] = 0;
    int hout[NC], hmb[NC];
#pragma unroll
    for (int k = 0; k < NC; k++) { hout[k] = 0; inp[k] = 0; mg[k] = 127; hmb[k] = (int)((mw[k >> 2] >> (8 * (k & 3))) & 0xffu); }
    // One ordered step per block of `block` rows. A step is a chain of dependent instructions of a single wave (the
    // next block reads what this one wrote), so its length is what a hazard layer costs: rel = jj - start is kept
    // incrementally (one subtract + one unsigned compare select the rows of the block).
    __builtin_amdgcn_s_setprio(3);
    bool lane_chain = false;
    if constexpr (NC == 2 && DEG <= kLaneChainMaxDeg && !PR) lane_chain = tab != nullptr; // wave-uniform (header bit 12)
    if constexpr (NC == 2 && DEG <= kLaneChainMaxDeg && !PR) if (lane_chain) {
        // LANE CHAIN (one hazard pair, block <= 64, host-ordered so that entry 0's bit of row r is entry 1's bit of
        // row r + block). A lone wave issues one instruction per ~6.7 cycles whatever it is, so an ordered step costs
        // its instruction count: the recurrence r -> r + block is walked by the `block` lanes that own rows
        // 0..block-1 with the chained LLR in a register, ~20 instructions per step, no exec-mask bookkeeping, no LDS
        // hand-over, no barrier per step; everything else happens before and after, in parallel over all rows:
        //   A  rows < block (chain heads, nothing precedes them): full two-entry step; entry-1 LLR written at once
        //      (rows >= 360 - block read it as their entry-0 LLR)                                        | barrier
        //   B  rows >= block: read entry 0 (no earlier row of this layer writes it), publish
        //      {inp0, partial min0, partial sign, message byte 1}                                        | barrier
        //   C  chain lanes: incoming entry-1 LLR -> new entry-0 LLR of row r, incoming value logged      | barrier
        //   D  rows >= block: complete both outputs from the logged value; write entry 1; the last row of a chain
        //      also writes entry 0 (in the reference's order it is the final writer of that bit).
        uint8_t* ulog = reinterpret_cast<uint8_t*>(tab + kM + block); // after the per-row records (360 rows + one block of padding)
        int chained = 0x80;
        const bool head = work && jj < block, body = work && jj >= block;
        if (head) {
            const int L0 = lds_rd(ad[0]), L1 = lds_rd(ad[1]);
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
            chained = sat_sum_u8(inp[0], hout[0]);
            lds_wr(ad[1], sat_sum_u8(inp[1], hout[1]));
        }
        __syncthreads();
        if (body) {
            const int L0 = lds_rd(ad[0]);
            inp[0] = min(max(L0 - hmb[0], -128), 127);
            mg[0] = mag_raw(L0, hmb[0]);
            tab[jj] = ((uint32_t)inp[0] & 0x1ffu) | ((uint32_t)min0 << 9) | (((uint32_t)signs >> 31) << 16) | ((uint32_t)hmb[1] << 24);
        }
        __syncthreads();
        if (head) {
            // rows past 359 read the padding of the table and log into the padding: no per-lane predicate in the loop
            const uint32_t* tp = tab + jj + block;
            uint8_t* up = ulog + jj + block;
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
        __syncthreads();
        if (body) {
            const int L1 = ulog[jj];
            inp[1] = min(max(L1 - hmb[1], -128), 127);
            mg[1] = mag_raw(L1, hmb[1]);
            int o0, o1;
            asm("v_med3_i32 %0, %1, 0, %2" : "=v"(o0) : "v"(mg[1]), "v"(min0));
            asm("v_med3_i32 %0, %1, 0, %2" : "=v"(o1) : "v"(mg[0]), "v"(min0));
            const int s0 = (signs ^ inp[1]) >> 31, s1 = (signs ^ inp[0]) >> 31;
            hout[0] = (o0 ^ s0) - s0;
            hout[1] = (o1 ^ s1) - s1;
            lds_wr(ad[1], sat_sum_u8(inp[1], hout[1]));
            if (jj + block >= kM) lds_wr(ad[0], sat_sum_u8(inp[0], hout[0]));
        }
    }
    int rel = (work && !lane_chain) ? jj : 0x40000000;
    for (int start = lane_chain ? kM : 0; start < kM; start += block, rel -= block) {
        if ((uint32_t)rel < (uint32_t)block) {
            if constexpr (NC == 2) {
                // two hazard entries: each one's magnitude sent back is min(partial min0, the other's magnitude) =
                // med3(raw other, 0, min0) (min0 is already clamped to [0, 126])
                const int L0 = lds_rd(ad[0]), L1 = lds_rd(ad[1]);
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
                lds_wr(ad[0], sat_sum_u8(inp[0], hout[0]));
                lds_wr(ad[1], sat_sum_u8(inp[1], hout[1]));
            } else {
            int Lh[NC];
#pragma unroll
            for (int k = 0; k < NC; k++) Lh[k] = lds_rd(ad[k]);
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
                lds_wr(ad[k], sat_sum_u8(inp[k], out));
            }
            }
        }
        // the next block reads what this one wrote: a workgroup barrier, unless both blocks sit inside one and the
        // same wavefront (LDS operations of a wave execute in program order)
        if ((start >> 6) != ((start + 2 * block - 1) >> 6)) __syncthreads();
    }
    __syncthreads();
    if constexpr (NC == 2) { mg[0] = clamp_mag(mg[0]); mg[1] = clamp_mag(mg[1]); } // raw in the loop (127 where no step ran: idle rows)
#pragma unroll
    for (int k = 0; k < NC; k++) {
        min1 = min(max(mg[k], min0), min1);
        min0 = min(min0, mg[k]);
        signs ^= inp[k];
        nm[k >> 2] |= (uint32_t)(min(max(hout[k], -32), 31) + 128) << (8 * (k & 3));
    }
    const int s01 = min0 + min1;
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
                else if (!(LAYER0 && k == DEG - 1) || last_valid) lds_wr(ad[k], nl);
                nm[k >> 2] |= (uint32_t)(min(max(out, -32), 31) + 128) << (8 * (k & 3));
            }
        }
        if (PR) nm[1] = (nm[1] & 0x00ffffffu) | ((uint32_t)spare << 24);
    }
}

// degrees DMAX-7 .. DMAX are instantiated for kernel variant DMAX
#define DVBS2_DEG_CASE(D) case D: if constexpr (D >= 3 && D <= DMAX && D > DMAX - 8) { \
        if (layer0) check_node<(D >= 3 ? D : 3), true>(lds_all, ent, jj, lb, mw, nm); else check_node<(D >= 3 ? D : 3), false>(lds_all, ent, jj, lb, mw, nm); } break;
#define DVBS2_DEG_SWITCH switch (deg) { \
        DVBS2_DEG_CASE(3) DVBS2_DEG_CASE(4) DVBS2_DEG_CASE(5) DVBS2_DEG_CASE(6) DVBS2_DEG_CASE(7) DVBS2_DEG_CASE(8) \
        DVBS2_DEG_CASE(9) DVBS2_DEG_CASE(10) DVBS2_DEG_CASE(11) DVBS2_DEG_CASE(12) DVBS2_DEG_CASE(13) DVBS2_DEG_CASE(14) \
        DVBS2_DEG_CASE(15) DVBS2_DEG_CASE(16) DVBS2_DEG_CASE(17) DVBS2_DEG_CASE(18) DVBS2_DEG_CASE(19) DVBS2_DEG_CASE(20) \
        DVBS2_DEG_CASE(21) DVBS2_DEG_CASE(22) DVBS2_DEG_CASE(23) DVBS2_DEG_CASE(24) DVBS2_DEG_CASE(25) DVBS2_DEG_CASE(26) \
        DVBS2_DEG_CASE(27) DVBS2_DEG_CASE(28) DVBS2_DEG_CASE(29) DVBS2_DEG_CASE(30) DVBS2_DEG_CASE(31) DVBS2_DEG_CASE(32) \
        default: break; }

#define DVBS2_HAZ_CALL(D, NCV) { if constexpr (D - 2 >= NCV) { \
        if (layer0) check_node_hazard<D, NCV, true>(lds_all, ent, jj, lb, work, block, mw, nm, 0, nullptr, htab); else check_node_hazard<D, NCV, false>(lds_all, ent, jj, lb, work, block, mw, nm, 0, nullptr, htab); } }
#define DVBS2_HAZ_CASE(D) case D: if constexpr (D >= 4 && D <= DMAX && D > DMAX - 8) { \
        if (nc == 2) DVBS2_HAZ_CALL((D >= 4 ? D : 4), 2) else if (nc == 4) DVBS2_HAZ_CALL((D >= 4 ? D : 4), 4) else DVBS2_HAZ_CALL((D >= 4 ? D : 4), 8) } break;
#define DVBS2_HAZ_SWITCH switch (deg) { \
        DVBS2_HAZ_CASE(4) DVBS2_HAZ_CASE(5) DVBS2_HAZ_CASE(6) DVBS2_HAZ_CASE(7) DVBS2_HAZ_CASE(8) \
        DVBS2_HAZ_CASE(9) DVBS2_HAZ_CASE(10) DVBS2_HAZ_CASE(11) DVBS2_HAZ_CASE(12) DVBS2_HAZ_CASE(13) DVBS2_HAZ_CASE(14) \
        DVBS2_HAZ_CASE(15) DVBS2_HAZ_CASE(16) DVBS2_HAZ_CASE(17) DVBS2_HAZ_CASE(18) DVBS2_HAZ_CASE(19) DVBS2_HAZ_CASE(20) \
        DVBS2_HAZ_CASE(21) DVBS2_HAZ_CASE(22) DVBS2_HAZ_CASE(23) DVBS2_HAZ_CASE(24) DVBS2_HAZ_CASE(25) DVBS2_HAZ_CASE(26) \
        DVBS2_HAZ_CASE(27) DVBS2_HAZ_CASE(28) DVBS2_HAZ_CASE(29) DVBS2_HAZ_CASE(30) DVBS2_HAZ_CASE(31) DVBS2_HAZ_CASE(32) \
        default: break; }

// MINW = 6 ("dense"): compiled for 80 VGPRs so that two pair-workgroups share a CU when the frames are short enough for
// LDS. It spills and only pays where ordered hazard steps dominate (ldpc_hip.hip picks it).
template <int DMAX, bool TIMING, int MINW = 1>
__global__ __launch_bounds__(kThreads, MINW) void ldpc_layered_kernel(
    const uint32_t* __restrict__ recs, const int8_t* __restrict__ llr_in, uint8_t* __restrict__ state,
    uint32_t* __restrict__ msgs, int* __restrict__ iters, int* __restrict__ good, const int* __restrict__ target,
    int n_frames, int N, int K, int q, int cap, int stop_on_good, unsigned long long* __restrict__ tdbg)
{
    unsigned long long tm_bar = 0, tm_body = 0, tm_conf = 0, tm_synd = 0, tm_sweep = 0, tm_load = 0, tm_s1 = 0;
#define TSTAMP(x) do { if (TIMING) { x = __builtin_readcyclecounter(); } } while (0)
    unsigned long long tA = 0, tB = 0, tC = 0, tS0 = 0, tS1 = 0;
    if (!llr_in) { // resume launch: a workgroup whose frames are both at their target leaves before touching LDS
        const int fa = 2 * (int)blockIdx.x, fb = fa + 1;
        const bool ta = fa < n_frames && iters[fa] < target[fa];
        const bool tb = fb < n_frames && iters[fb] < target[fb];
        if (!ta && !tb) return;
    }
    TSTAMP(tA);
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_all[];
    constexpr int RS = rec_stride(DMAX);
    constexpr int MW = DMAX / 4; // message dwords per check (fixed per kernel variant)
    const int half = __builtin_amdgcn_readfirstlane(threadIdx.x >= kHalf ? 1 : 0); // wave-uniform, and the compiler knows it
    const int tid = threadIdx.x - half * kHalf;
    const int lb_rel = half * (int)half_lds_bytes(N);
    const int lb = lb_rel + lds_address_of(lds_all); // absolute LDS address of this frame's region
    uint8_t* lds = lds_all + lb_rel;
    uint32_t* sv = reinterpret_cast<uint32_t*>(lds + N); // N % 8 == 0
    volatile int* flags = reinterpret_cast<volatile int*>(sv + (N / kM) * kSvWords); // [0] bad-or, [1] finished, [2] pre-test failed, [3] full test needed
    volatile int* other_flags = reinterpret_cast<volatile int*>(
        lds_all + (1 - half) * half_lds_bytes(N) + N + (size_t)(N / kM) * kSvWords * 4);
    const int f = 2 * blockIdx.x + half;
    const bool have_frame = f < n_frames;
    const int lane = tid & 63, wave = tid >> 6;
    const int NG = N / kM;
    const bool active = tid < kM;

    int it = 0, tgt = 0;
    bool finished = !have_frame; // this half has nothing (more) to do; it still takes part in every barrier
    if (have_frame) {
        tgt = target ? target[f] : cap;
        if (llr_in) {
            const uint2* src = reinterpret_cast<const uint2*>(llr_in + (size_t)f * N);
            for (int c = tid; c < N / 8; c += kHalf) {
                uint2 v = src[c];
                v.x ^= 0x80808080u; v.y ^= 0x80808080u;
                const int n = 8 * c;
                if (n < K) *reinterpret_cast<uint2*>(lds + n) = v; // K % 8 == 0
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
                for (int c = tid; c < N / 8; c += kHalf) *reinterpret_cast<uint2*>(lds + 8 * c) = src[c];
            }
        }
    }
    const bool untouched = finished; // never loaded: must not write state/iters/good back
    if (tid == 0) { flags[0] = 0; flags[2] = 0; flags[3] = 0; flags[1] = finished ? 1 : 0; }
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
    // bnl = 0 before the first update (layered_decoder.hh:27-31,149): a frame's first sweep (it == 0, in the first pass or
    // when a frame that stopped at once is resumed) takes offset-binary zero bytes instead of loading them -- no memset
    // of the record area, no read traffic in sweep 0
    bool is_good = false;

    for (;;) {
        TSTAMP(tS0);
        // ---- syndrome test (layered_decoder.hh:32-49, algorithms.hh:195-202): bad if any check has a zero
        // LLR or an odd number of negative LLRs. Barriers are taken by every thread; work only by halves that need it.
        const bool need_synd = !finished && (stop_on_good || it >= tgt);
        // Pre-test (the reference's bad() also returns at the first failing check): the 360 checks of ONE layer,
        // tested edge by edge. A failure here is final; only a frame that passes pays for the full test below.
        if (need_synd && active) {
            const int i0 = it % q;
            const uint32_t* rec = recs + (size_t)i0 * RS;
            const int deg = (int)(rec[0] & 0xffu) + 2;
            uint32_t x = 0, z = 0;
            for (int k = 0; k < deg; k++) {
                const int a0 = tid + (int)rec[4 + 2 * k] - ((uint32_t)tid < rec[5 + 2 * k] ? 0 : kM);
                uint32_t v = lds[a0];
                if (i0 == 0 && k == deg - 1 && tid == 0) v = 0x81u; // check (0,0) has no previous parity: neutral +1
                x ^= v;
                z |= (v == 0x80u);
            }
            // offset binary: the sign bit is inverted, so the count of negatives is deg - popcount(bit 7)
            const int bad_pre = (int)((((x >> 7) ^ (uint32_t)deg) & 1u) | z);
            if (__ballot(bad_pre) != 0 && lane == 0) flags[2] = 1;
        }
        __syncthreads();
        const bool need_full = need_synd && flags[2] == 0;
        if (tid == 0) flags[3] = need_full ? 1 : 0;
        __syncthreads();
        const bool full_any = flags[3] != 0 || other_flags[3] != 0; // uniform over the workgroup
        if (full_any) {
        if (need_full) {
            // Step 1: 360-bit sign vector per group via wave ballots (4 groups per trip to batch the LDS reads).
            unsigned long long zero_any = 0;
            for (int g0 = 0; g0 < NG; g0 += 4) {
                uint32_t v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = (active && g0 + u < NG) ? lds[kM * (g0 + u) + tid] : 0xffu;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const unsigned long long neg = __ballot(v[u] < 0x80u);
                    zero_any |= __ballot(v[u] == 0x80u);
                    if (lane == 0 && g0 + u < NG) *reinterpret_cast<uint2*>(&sv[(g0 + u) * kSvWords + 2 * wave]) = make_uint2((uint32_t)neg, (uint32_t)(neg >> 32));
                }
            }
            if (zero_any != 0 && lane == 0) flags[0] = 1;
        }
        TSTAMP(tC); tm_s1 += tC - tS0;
        __syncthreads();
        if (need_full && tid < NG) { // wrap extension: bits 360+u = bit u
            uint32_t* p = sv + tid * kSvWords;
            const uint32_t w0 = p[0], w1 = p[1];
            p[11] = (p[11] & 0xffu) | (w0 << 8);
            p[12] = (w0 >> 24) | (w1 << 8);
        }
        __syncthreads();
        if (need_full) {
            // Step 2: parity word (layer i, lanes 32w..32w+31) = xor over entries of the rotated sign vectors
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
                        const uint32_t* p = sv + (g360 / kM) * kSvWords + (t0 >> 5);
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
        __syncthreads();
        } // full_any
        if (need_synd) is_good = need_full && flags[0] == 0;
        if (!finished && (it >= tgt || (stop_on_good && is_good))) finished = true;
        __syncthreads(); // everyone has read flags[0]
        if (tid == 0) { flags[0] = 0; flags[2] = 0; flags[1] = finished ? 1 : 0; }
        __syncthreads();
        TSTAMP(tS1); tm_synd += tS1 - tS0;
        if (finished && other_flags[1]) break; // uniform over the workgroup

        // ---- one update sweep: layered_decoder.hh:50-79 ----
        // Threads 360..383 of a half mirror check row 359: same reads, same results, same (duplicate) writes. That
        // keeps the whole sweep free of per-lane predicates: `work` is wave-uniform.
        const bool work = !finished;
        const int row = tid < kM ? tid : kM - 1;
        const int row4 = row * 4;
        const bool zero_msgs = it == 0; // uniform over the half
        uint32_t pre[MW]; // messages of the next layer for check tid, loaded one layer ahead
        if (work) {
#pragma unroll
            for (int w = 0; w < MW; w++) pre[w] = zero_msgs ? 0x80808080u : MSG_LD(0, w, row4);
        }
        // Layer records are double-buffered in SGPRs: the scalar loads of layer i+1 are issued at the top of layer i
        // (an un-prefetched s_load at the head of every layer was a quarter of the sweep time). Small records are
        // buffered whole; for the large ones only every 8th dword is carried over -- enough to pull each cache
        // line of the next record into the scalar cache -- and the rest is loaded at the top of the layer.
        constexpr int PF = DMAX <= 12 ? 1 : 8;
        uint32_t nhdr = recs[0];
        uint32_t nent[2 * DMAX];
#pragma unroll
        for (int k = 0; k < 2 * DMAX; k += PF) nent[k] = recs[4 + k];
        for (int i = 0; i < q; i++) {
            const uint32_t hdr = nhdr;
            uint32_t ent[2 * DMAX];
#pragma unroll
            for (int k = 0; k < 2 * DMAX; k++) ent[k] = (k % PF == 0) ? nent[k] : recs[(size_t)i * RS + 4 + k];
            {
                const uint32_t* nrec = recs + (size_t)(i + 1 < q ? i + 1 : 0) * RS;
                nhdr = nrec[0];
#pragma unroll
                for (int k = 0; k < 2 * DMAX; k += PF) nent[k] = nrec[4 + k];
            }
            const int deg = (int)(hdr & 0xffu) + 2;
            const int nc = (int)((hdr >> 8) & 0xfu);
            uint32_t* htab = ((hdr >> 12) & 1u) ? sv : nullptr; // lane-chain scratch: the sign-vector area is idle during a sweep
            const int block = (int)(hdr >> 16);
            const bool layer0 = (i == 0);
            const int mso = i * kLayerBytes; // scalar byte offset of this layer's message records
            TSTAMP(tA);
            if (hdr & 0x8000u) __syncthreads();
            TSTAMP(tB); tm_bar += tB - tA;
            if (block >= kM) {
                // regular layer: all 360 checks at once
                if (work) {
                    const int jj = row;
                    uint32_t mw[MW], nm[MW];
#pragma unroll
                    for (int w = 0; w < MW; w++) mw[w] = pre[w];
                    if (i + 1 < q && !zero_msgs) {
#pragma unroll
                        for (int w = 0; w < MW; w++) pre[w] = MSG_LD(mso + kLayerBytes, w, row4);
                    }
                    DVBS2_DEG_SWITCH
#pragma unroll
                    for (int w = 0; w < MW; w++) MSG_ST(nm[w], mso, w, row4);
                }
                TSTAMP(tC); tm_body += tC - tB;
            } else {
                if (nc != kHazardWalk) {
                    // sequential-order hazard inside the layer: check_node_hazard (every thread takes every barrier)
                    const int jj = row;
                    uint32_t mw[MW], nm[MW];
#pragma unroll
                    for (int w = 0; w < MW; w++) mw[w] = work ? pre[w] : 0x80808080u;
                    if (work && i + 1 < q && !zero_msgs) {
#pragma unroll
                        for (int w = 0; w < MW; w++) pre[w] = MSG_LD(mso + kLayerBytes, w, row4);
                    }
                    DVBS2_HAZ_SWITCH
                    if (work) {
#pragma unroll
                        for (int w = 0; w < MW; w++) MSG_ST(nm[w], mso, w, row4);
                    }
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
                    __syncthreads();
                    if (work && i + 1 < q && !zero_msgs) {
#pragma unroll
                        for (int w = 0; w < MW; w++) pre[w] = MSG_LD(mso + kLayerBytes, w, row4);
                    }
                }
                TSTAMP(tC); tm_conf += tC - tB;
            }
        }
        TSTAMP(tA);
        __syncthreads();
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
        for (int c = tid; c < N / 8; c += kHalf) dst[c] = *reinterpret_cast<const uint2*>(lds + 8 * c);
    }
}


// ---- host-side launch interface of one kernel variant (defined in ldpc_inst_*.hip) ----
struct LdpcLaunch {
    const uint32_t* recs; const int8_t* llr_in; uint8_t* state; uint32_t* msgs; int* iters; int* good; const int* target;
    int n_frames, N, K, q, cap, stop_on_good; unsigned long long* tdbg;
    size_t lds_bytes; hipStream_t stream;
    bool dense; // the 80-VGPR build of the kernel (two workgroups per CU), see kDenseBuilt
};
template <int DMAX> hipError_t ldpc_variant_prepare(size_t lds_bytes);
template <int DMAX> void ldpc_variant_launch(const LdpcLaunch& a);

#ifdef DVBS2_LDPC_INSTANTIATE
// The cycle-stamped variant (DVBS2_TIMING=1, tools/exp_tables.py) is only built for DMAX = 8 -- the headline tables --
// to keep the build time of the large variants down; elsewhere the request is ignored.
template <int DMAX> constexpr bool kTimingBuilt = (DMAX == 8);
// only the degree class 5..12 survives 80 VGPRs (120 B of scratch); the classes of short 5/6 and 8/9 (DMAX 20, 28) spill so
// much that they run 8x slower (measured)
template <int DMAX> constexpr bool kDenseBuilt = (DMAX == 12);
template <int DMAX> hipError_t ldpc_variant_prepare(size_t lds_bytes)
{
    hipError_t e = hipFuncSetAttribute((const void*)ldpc_layered_kernel<DMAX, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e != hipSuccess) return e;
    if constexpr (kDenseBuilt<DMAX>) {
        e = hipFuncSetAttribute((const void*)ldpc_layered_kernel<DMAX, false, 6>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
        if (e != hipSuccess) return e;
    }
    if constexpr (kTimingBuilt<DMAX>)
        return hipFuncSetAttribute((const void*)ldpc_layered_kernel<DMAX, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    return hipSuccess;
}
template <int DMAX> void ldpc_variant_launch(const LdpcLaunch& a)
{
    const dim3 grid((a.n_frames + 1) / 2), block(kThreads);
    if constexpr (kTimingBuilt<DMAX>) {
        if (a.tdbg) {
            hipLaunchKernelGGL((ldpc_layered_kernel<DMAX, true>), grid, block, a.lds_bytes, a.stream, a.recs, a.llr_in, a.state, a.msgs,
                               a.iters, a.good, a.target, a.n_frames, a.N, a.K, a.q, a.cap, a.stop_on_good, a.tdbg);
            return;
        }
    }
    if constexpr (kDenseBuilt<DMAX>) {
        if (a.dense) {
            hipLaunchKernelGGL((ldpc_layered_kernel<DMAX, false, 6>), grid, block, a.lds_bytes, a.stream, a.recs, a.llr_in, a.state, a.msgs,
                               a.iters, a.good, a.target, a.n_frames, a.N, a.K, a.q, a.cap, a.stop_on_good, nullptr);
            return;
        }
    }
    hipLaunchKernelGGL((ldpc_layered_kernel<DMAX, false>), grid, block, a.lds_bytes, a.stream, a.recs, a.llr_in, a.state, a.msgs,
                       a.iters, a.good, a.target, a.n_frames, a.N, a.K, a.q, a.cap, a.stop_on_good, nullptr);
}
template hipError_t ldpc_variant_prepare<DVBS2_LDPC_INSTANTIATE>(size_t);
template void ldpc_variant_launch<DVBS2_LDPC_INSTANTIATE>(const LdpcLaunch&);
#endif

} // namespace dvbs2
