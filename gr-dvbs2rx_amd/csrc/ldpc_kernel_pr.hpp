// ldpc_kernel_pr.hpp -- "parity in records" variant of the layered LDPC kernel for low-rate normal frames.
//
// Why: a frame running alone on a CU is latency bound (about 250 in-order instructions per wave per layer plus an
// LDS round trip and a barrier): one pair of frames takes 1.36x the time of one frame, and going from one pair
// to two pairs per CU gained 1.45-1.5x on short frames. What limits normal frames to one pair per CU is LDS:
// 64800 LLR bytes per frame. But parity row i is only ever touched by thread j of layers i and i+1 (own parity,
// then previous parity), so it can live in registers between those two layers and in byte 7 of the per-thread
// message record in between sweeps (ldpc_kernel.hpp, check_node<.., PR>). LDS then holds K + 360 bytes per frame
// (information LLRs + parity row q-1) and, for K <= 32400 (rates up to 1/2) with check degree <= 7, TWO pair
// workgroups fit a CU: 24 waves, 6 per SIMD, at 80 VGPRs.
//
// Layout per workgroup: [half 0 LLRs][half 1 LLRs][sign vectors, shared by the halves][flags 0][flags 1].
// Message record of (layer i, row j): 2 dwords = message bytes 0..deg-1, byte 7 = parity LLR P[i-1][j] as left by
// layer i (offset binary); record 0 byte 7 is unused.
// W1 (tables of check degree <= 4): ONE dword per record -- four messages as 6-bit fields (the stored message is clamped to
// [-32, 31], R7: value + 32 in bits 6k .. 6k+5) and the parity LLR in byte 3. These tables move 8 bytes per check and layer each way
// for 4-5 useful ones and run at ~4.9 TB/s of real traffic (short 1/4: 2.3 x the algorithmic bytes): halving the record is worth
// the ~18 VALU instructions of the conversion around the node.
#pragma once
#include "ldpc_kernel.hpp"
#include <cstdio>
#include <cstdlib>

#ifndef DVBS2_PR_DEFER_STORE
#define DVBS2_PR_DEFER_STORE 1
#endif
// Round 6: the degree-3 / 4 node of the one-dword-record kernel takes the minimum over the OTHER links directly (see check_node_pr6). Interleaved A/B against
// "two smallest + select" (three repetitions, gpurun_out/r6m): short 1/4 1059.8 -> 1074.1 k (+1.3 %), S2X short 1220 -> 1232 k, medium 1/5 and 11/45 +1.0 / +1.1 %,
// 1/4 normal and S2X 2/9 normal +0.9 / +1.2 %; the kernel's other tables (two-dword records) 1.000 / 1.001.
// Round 6: the record accesses of the layer loop as (uniform pointer of the layer) + (per-lane byte offset, loop invariant), and `finished` passed through
// v_readfirstlane (it is read from LDS, so the compiler took it -- and with it the index of the deferred store -- for divergent): the two v_mad_u64_u32 per layer
// become scalar multiplies and one v_lshl_add_u64 each, the loop's own VALU instructions drop to nine per layer. Interleaved A/B x 3 (notes/r06_stamps/pr_scalar_base_ab.txt):
// short 1/4 1070.3 -> 1132.3 k (+5.8 %), S2X short +5.7 %, medium 11/45 +4.8 %, 1/4 normal +5.2 %; two-dword records +0.6 ... +1.4 %. (The MUBUF form of the same idea,
// a buffer descriptor per access, had LOST 2-4 % earlier in the round.)

namespace dvbs2 {

__host__ __device__ constexpr size_t pr_half_bytes(int K) { return ((size_t)K + kM + 15) / 16 * 16; }
// two frames + sign-vector areas: one PER FRAME (round 5: the frames run their full syndrome tests at the same time), or ONE shared by the
// workgroup where two do not fit twice into the 160 KB of a CU (normal frames forced onto this kernel: the frames then take turns; bit 3 of the flag word)
__host__ __device__ constexpr size_t pr_lds_bytes(int N, int K, bool shared_sv = false) { return 2 * pr_half_bytes(K) + (shared_sv ? 1 : 2) * (size_t)(N / kM) * kSvWords * 4 + 64; }

#ifdef DVBS2_LDPC_INSTANTIATE_PR
#define DVBS2_PR_CASE(D) case D: { \
        if (first_layer) check_node<D, true, true, false>(lds_all, ent, jj, lb, mw, nm, own_in, &carry); \
        else if (last_layer) check_node<D, false, true, true>(lds_all, ent, jj, lb, mw, nm, own_in, &carry); \
        else check_node<D, false, true, false>(lds_all, ent, jj, lb, mw, nm, own_in, &carry); } break;
#define DVBS2_PR_SWITCH switch (deg) { DVBS2_PR_CASE(3) DVBS2_PR_CASE(4) DVBS2_PR_CASE(5) DVBS2_PR_CASE(6) DVBS2_PR_CASE(7) default: break; }
#define DVBS2_PRH_CALL(D, NCV) { if constexpr (D - 2 >= NCV) { \
        if (first_layer) check_node_hazard<D, NCV, true, true, false>(lds_all, ent, jj, lb, work, block, 0, mw, nm, own_in, &carry, nullptr, nullptr, pr_epoch, 0); \
        else if (last_layer) check_node_hazard<D, NCV, false, true, true>(lds_all, ent, jj, lb, work, block, 0, mw, nm, own_in, &carry, nullptr, nullptr, pr_epoch, 0); \
        else check_node_hazard<D, NCV, false, true, false>(lds_all, ent, jj, lb, work, block, 0, mw, nm, own_in, &carry, nullptr, nullptr, pr_epoch, 0); } }
#define DVBS2_PRH_CASE(D) case D: { if (nc == 2) DVBS2_PRH_CALL(D, 2) else if (nc == 4) DVBS2_PRH_CALL(D, 4) } break;
#define DVBS2_PRH_SWITCH switch (deg) { DVBS2_PRH_CASE(4) DVBS2_PRH_CASE(5) DVBS2_PRH_CASE(6) DVBS2_PRH_CASE(7) default: break; }

// 6-bit record <-> the byte format the check nodes take (mw[0] = four offset-binary message bytes, byte 3 of mw[1] = parity LLR)
__device__ __forceinline__ void pr_w1_expand(uint32_t x, uint32_t* mw)
{
    const uint32_t f0 = x & 0x3fu, f1 = (x >> 6) & 0x3fu, f2 = (x >> 12) & 0x3fu, f3 = (x >> 18) & 0x3fu;
    mw[0] = ((f0 | (f1 << 8)) | ((f2 | (f3 << 8)) << 16)) + 0x60606060u; // field = m + 32, byte = m + 128
    mw[1] = (x & 0xff000000u) | 0x00808080u;
}
__device__ __forceinline__ uint32_t pr_w1_compress(const uint32_t* nm)
{
    // byte = m + 128 with m in [-32, 31] (R7), i.e. 0x60 .. 0x9f: field = m + 32 = (byte & 0x3f) ^ 0x20 (no borrow between bytes; a
    // byte the node left unset, 0x00 or 0x80, becomes the zero message)
    const uint32_t y = (nm[0] & 0x3f3f3f3fu) ^ 0x20202020u;
    return (y & 0x3fu) | ((y >> 2) & 0xfc0u) | ((y >> 4) & 0x3f000u) | ((y >> 6) & 0xfc0000u) | (nm[1] & 0xff000000u);
}

// check_node<DEG, LAYER0, PR = true, LAST> (ldpc_kernel.hpp) on the one-dword record itself: messages come out of and go back into
// the 6-bit fields without the detour through byte words (two instructions per message each way instead of ~4.5).
__device__ __forceinline__ uint32_t vmin3_u32(uint32_t a, uint32_t b, uint32_t c) { uint32_t r; asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(c)); return r; }
__device__ __forceinline__ uint32_t usub_sat1(uint32_t a) { uint32_t r; asm("v_sub_u32_e64 %0, %1, 1 clamp" : "=v"(r) : "v"(a)); return r; } // max(a - 1, 0)
template <int DEG, bool LAYER0, bool LAST>
__device__ __forceinline__ uint32_t check_node_pr6(uint8_t* __restrict__ lds, const uint32_t* ent, int jj, int lb, uint32_t x /*this layer's record*/,
                                                   int own_in, int* carry)
{
    static_assert(DEG >= 3 && DEG <= 4, "one-dword records hold four messages");
    constexpr bool OWN_REG = !LAST, PREV_REG = !LAYER0;
    __builtin_amdgcn_s_setprio(0);
    int ad[DEG], Lb[DEG];
    const int jjb = jj + lb, jjb360 = jjb - kM;
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        if (k >= DEG - 2 && !(LAYER0 && k == DEG - 1)) ad[k] = jjb + (int)ent[2 * k];
        else ad[k] = wrap_addr(jj, jjb, jjb360, ent[2 * k], ent[2 * k + 1]);
    }
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        if (OWN_REG && k == DEG - 2) Lb[k] = own_in;
        else if (PREV_REG && k == DEG - 1) Lb[k] = *carry;
        else Lb[k] = lds_rd(ad[k]);
    }
    const bool last_valid = !LAYER0 || jj != 0;
    int spare = 0x80;
    int inp[DEG], mg[DEG];
    int signs = 0;
#pragma unroll
    for (int k = 0; k < DEG; k++) {
        const int mb = (int)((x >> (6 * k)) & 0x3fu) + 96; // field = m + 32 -> offset-binary byte m + 128
        int d = min(max(Lb[k] - mb, -128), 127);
        int mag = (int)__builtin_amdgcn_sad_u16((uint32_t)Lb[k], (uint32_t)mb, 0u); // |Lb - mb| in [0, 255]
        if (LAYER0 && k == DEG - 1) { d = last_valid ? d : 0; mag = last_valid ? mag : kMagAbsent; }
        inp[k] = d; mg[k] = mag;
        signs ^= d;
    }
    __builtin_amdgcn_s_setprio(1);
    // Degree 3 / 4: the minimum over the OTHER links directly (R3 + R5's selection in one): v_min3_u32 of the other raw |Lb - mb| and 127
    // (R2's qabs bound), then one saturating subtract of the offset -- 3 half-rate + 3 full-rate instructions at degree 3 and 6 + 4 at degree 4,
    // against 7 + 4 and 10 + 5 for "two smallest, clamp both, select per link".
    int oth[DEG];
    if constexpr (DEG == 3) {
        oth[0] = (int)vmin3_u32((uint32_t)mg[1], (uint32_t)mg[2], 127u);
        oth[1] = (int)vmin3_u32((uint32_t)mg[0], (uint32_t)mg[2], 127u);
        oth[2] = (int)vmin3_u32((uint32_t)mg[0], (uint32_t)mg[1], 127u);
    } else {
        const uint32_t m01 = min((uint32_t)mg[0], (uint32_t)mg[1]), m23 = min((uint32_t)mg[2], (uint32_t)mg[3]);
        oth[0] = (int)vmin3_u32((uint32_t)mg[1], m23, 127u);
        oth[1] = (int)vmin3_u32((uint32_t)mg[0], m23, 127u);
        oth[2] = (int)vmin3_u32((uint32_t)mg[3], m01, 127u);
        oth[3] = (int)vmin3_u32((uint32_t)mg[2], m01, 127u);
    }
    uint32_t y = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        if (k < DEG) {
            const int other = (int)usub_sat1((uint32_t)oth[k]);
            const int sg = (signs ^ inp[k]) >> 31;
            const int out = (other ^ sg) - sg;
            const int nl = sat_sum_u8(inp[k], out);
            if (OWN_REG && k == DEG - 2) *carry = nl;
            else if (PREV_REG && k == DEG - 1) spare = nl;
            else if (!(LAYER0 && k == DEG - 1) || last_valid) lds_wr(ad[k], nl);
            const uint32_t f = (uint32_t)(min(max(out, -32), 31) + 32); // R7, as a field
            y = k == 0 ? f : (f << (6 * k)) | y;
        } else y |= 32u << (6 * k); // no such link: the zero message
    }
    __builtin_amdgcn_s_setprio(3);
    return y | ((uint32_t)spare << 24);
}

// check_node_v2 (ldpc_kernel.hpp: pairs of edges in the 16-bit halves of a register, one-add addresses from the wave's record) for the regular middle
// layers of the two-dword-record kernel (round 6, last session). Differences: the two parity entries DEG-2 (own) and DEG-1 (previous) come from and go
// back to registers (own_in / carry / byte 3 of word 1), LDS holds offset-binary bytes (the plain nodes of layer 0, the last layer and the hazard layers
// share it), and word 1's byte 3 is the parity LLR -- at degree 7 that is where the pad half of pair 3 would keep its (always zero) message, so the pair is
// unpacked without it. Messages of such a (layer, wave) are two's complement bytes in pair order (private to the layer, as in the classic kernel).
template <int DEG>
__device__ __forceinline__ void check_node_v2_pr(const uint32_t* ent /*record words 4..: window offsets of the data entries [8], then (mask lo, mask hi)[2]*/,
                                                 int jjb, const uint32_t* mw, uint32_t* nm, int own_in, int* carry)
{
    static_assert(DEG >= 5 && DEG <= 7, "two-dword records of degree 5 .. 7");
    constexpr int DMAX = 8, ND = DEG - 2, NP = (DEG + 1) / 2;
    constexpr int NFIX = v2_nfix(DMAX) < ND ? v2_nfix(DMAX) : ND;
    constexpr bool ODD = (DEG & 1) != 0;
    __builtin_amdgcn_s_setprio(0);
    int ad[ND];
#pragma unroll
    for (int k = 0; k < ND; k++) ad[k] = jjb + (int)ent[k];
#pragma unroll
    for (int k = 0; k < NFIX; k++) ad[k] = fix_wrap(ad[k], ent[DMAX + 2 * k], ent[DMAX + 2 * k + 1]);
    int Lb[DEG];
#pragma unroll
    for (int k = 0; k < ND; k++) Lb[k] = lds_rd(ad[k]);
    Lb[DEG - 2] = own_in; Lb[DEG - 1] = *carry;
    v2s16 d[NP], a[NP];
    uint32_t sx = 0;
#pragma unroll
    for (int j = 0; j < NP; j++) {
        uint32_t M = msg_pair16<false>(mw, j);
        if (ODD && j == NP - 1 && (j & 1)) M &= 0x0000ff00u; // degree 7: byte 3 of word 1 is the parity LLR, not the pad's message
        const uint32_t hi = (ODD && j == NP - 1) ? 0x80u : (uint32_t)Lb[2 * j + 1];
        const uint32_t L = __builtin_amdgcn_perm(hi, (uint32_t)Lb[2 * j], 0x040c000cu) ^ kObPair<false>;
        d[j] = __builtin_elementwise_sub_sat(as_v2s(L), as_v2s(M));
        sx ^= as_u32(d[j]);
        a[j] = __builtin_elementwise_max(d[j], __builtin_elementwise_sub_sat(as_v2s(0u), d[j]));
    }
    __builtin_amdgcn_s_setprio(1);
    int mg[DEG];
#pragma unroll
    for (int k = 0; k < DEG; k++) mg[k] = (k & 1) ? (int)(as_u32(a[k >> 1]) >> 16) : (int)(as_u32(a[k >> 1]) & 0xffffu);
    int n0, n1;
    two_smallest<DEG>(mg, n0, n1);
    n0 &= 0x7f00; n1 &= 0x7f00;
    const int n0m = (int)__builtin_elementwise_sub_sat((uint32_t)n0, 256u), n1m = (int)__builtin_elementwise_sub_sat((uint32_t)n1, 256u);
    const int B0 = n0, B1 = n0 + n1m - n0m, T = n1m + n0;
    const v2s16 B0p = { (short)B0, (short)B0 }, B1p = { (short)B1, (short)B1 }, Tp = { (short)T, (short)T };
    const uint32_t tm = (uint32_t)((int)(sx ^ (sx << 16)) >> 31);
    uint32_t R[NP];
    int spare = 0x80;
#pragma unroll
    for (int j = 0; j < NP; j++) {
        const v2s16 c = __builtin_elementwise_min(__builtin_elementwise_max(a[j], B0p), B1p);
        const v2s16 other = Tp - c;
        const v2s16 sg = as_v2s(as_u32(d[j]) ^ tm) >> (v2s16){ 15, 15 };
        const v2s16 out = as_v2s(as_u32(other) ^ as_u32(sg)) - sg;
        const uint32_t nl = (as_u32(__builtin_elementwise_add_sat(d[j], out)) ^ kObPair<false>) >> 8; // bits 7:0 / 23:16: the new LLR bytes of the pair
        constexpr int kOwn = DEG - 2, kPrev = DEG - 1;
        const int e0 = 2 * j, e1 = 2 * j + 1;
        if (e0 < ND) lds_wr(ad[e0 < ND ? e0 : 0], (int)nl);
        else if (e0 == kOwn) *carry = (int)(nl & 0xffu);
        else if (e0 == kPrev) spare = (int)(nl & 0xffu);
        if (!(ODD && j == NP - 1)) {
            if (e1 < ND) lds_wr_hi(ad[e1 < ND ? e1 : 0], nl);
            else if (e1 == kOwn) *carry = (int)((nl >> 16) & 0xffu);
            else if (e1 == kPrev) spare = (int)((nl >> 16) & 0xffu);
        }
        R[j] = as_u32(__builtin_elementwise_min(__builtin_elementwise_max(out, (v2s16){ -32 * 256, -32 * 256 }), (v2s16){ 31 * 256, 31 * 256 }));
    }
    __builtin_amdgcn_s_setprio(3);
    if (ODD) R[NP - 1] &= 0x0000ffffu;
    msg_pack16<false, NP, 2>(R, nm);
    nm[1] = (nm[1] & 0x00ffffffu) | ((uint32_t)spare << 24);
}

// Step 1 of the full syndrome test of the parity-in-records kernel: sign bits / zero test of all N LLRs of one frame, 360 per group, into the
// frame's sign-vector area. A function of its own, NOT inlined: the full test runs rarely on never-converging input, and inlined its chunk of
// loads cost the sweep of every table 1-3 % through register allocation (round 5; the same rule as syndrome_sign_vectors in ldpc_kernel.hpp).
template <bool W1>
__device__ __attribute__((noinline)) unsigned long long pr_sign_vectors(const lds_byte_t* lds, lds_u32_t* sv, const uint32_t* __restrict__ msg_base,
                                                                        int K, int N, int q, int tid)
{
    constexpr int RW = W1 ? 1 : 2, PW = RW - 1;
    const int lane = tid & 63, wave = tid >> 6;
    const int NGD = K / kM, NG = N / kM;
    const bool active = tid < kM;
    unsigned long long zero_any = 0;
    auto put_row = [&](int g, uint32_t v) { // sign bits / zero test of the 360 LLRs of group g, one per lane
        const unsigned long long neg = __ballot(v < 0x80u);
        zero_any |= __ballot(v == 0x80u);
        if (lane == 0) *reinterpret_cast<lds_v2u_t*>(&sv[g * kSvWords + 2 * wave]) = (v2u32){ (uint32_t)neg, (uint32_t)(neg >> 32) };
    };
    for (int g = 0; g < NGD; g++) put_row(g, active ? (uint32_t)lds[kM * g + tid] : 0xffu); // information part: LDS
    put_row(NG - 1, active ? (uint32_t)lds[K + tid] : 0xffu);                              // parity row q - 1: LDS
    // Parity rows 0 .. q-2 live in the message records in HBM (byte 3 of word PW of layer r + 1). Round 5: kSynChunk of them are
    // requested at once, unconditionally (every thread of the half has a slot in a record word) -- one load per row inside the
    // per-row branch was one round trip to memory per row, 8.6 us of a 15.7 us full test of short 1/4.
    constexpr int kSynChunk = 16;
    for (int r0 = 0; r0 < q - 1; r0 += kSynChunk) {
        uint32_t w[kSynChunk];
#pragma unroll
        for (int c = 0; c < kSynChunk; c++) {
            w[c] = msg_base[((min(r0 + c, q - 2) + 1) * RW + PW) * kMsgStride + tid];
        }
#pragma unroll
        for (int c = 0; c < kSynChunk; c++) if (r0 + c < q - 1) put_row(NGD + r0 + c, active ? w[c] >> 24 : 0xffu);
    }
    return zero_any;
}

// Records use the PR LDS layout: data entries as in the classic kernel; own parity of the last layer at K + j;
// previous parity of layer 0 at K + (j + 359) mod 360 (S0 = K + 359, thr = 1).
// record word at (uniform word pointer) + (per-lane BYTE offset, loop invariant): the form the global instructions take a scalar base for
__device__ __forceinline__ uint32_t pr_ld(const uint32_t* sbase, uint32_t boff) { return *reinterpret_cast<const uint32_t*>(reinterpret_cast<const char*>(sbase) + boff); }
__device__ __forceinline__ void pr_st(uint32_t* sbase, uint32_t boff, uint32_t v) { *reinterpret_cast<uint32_t*>(reinterpret_cast<char*>(sbase) + boff) = v; }
template <bool W1, bool V2 = false /*packed nodes in the regular middle layers (per-wave records, two-dword records only)*/>
__global__ __launch_bounds__(kThreads, 6) void ldpc_layered_pr_kernel(
    const uint32_t* __restrict__ recs, const uint32_t* __restrict__ wrecs, const int8_t* __restrict__ llr_in, uint8_t* __restrict__ state,
    uint32_t* __restrict__ msgs, int* __restrict__ iters, int* __restrict__ good, const int* __restrict__ target,
    int n_frames, int N, int K, int q, int cap, int stop_on_good /*bit 0: stop at a good syndrome, bit 2: group-synchronous stop (ldpc_kernel.hpp, group_decide), bit 3: one sign-vector area per workgroup*/)
{
    if (!llr_in) { // resume launch: a workgroup whose frames are both at their target leaves before touching LDS
        const int fa = 2 * (int)blockIdx.x, fb = fa + 1;
        const bool ta = fa < n_frames && iters[fa] < target[fa];
        const bool tb = fb < n_frames && iters[fb] < target[fb];
        if (!ta && !tb) return;
    }
    extern __shared__ __attribute__((aligned(16))) uint8_t lds_all[];
    static_assert(!(W1 && V2), "packed nodes: two-dword records");
    constexpr int DMAX = 8, RS = rec_stride(DMAX), MW = 2 /*words the check nodes take*/, RW = W1 ? 1 : 2 /*words per stored record*/;
    constexpr int RSW = rec_stride_wave(DMAX);
    constexpr int LS = V2 ? 6 * RSW : RS; // dwords from one layer's sweep record to the next (V2: this wave's records, wrecs[(layer * 6 + wave) * RSW])
    constexpr int PW = RW - 1; // word of a record whose top byte is the parity LLR
    const int half = __builtin_amdgcn_readfirstlane(threadIdx.x >= kHalf ? 1 : 0);
    const int tid = threadIdx.x - half * kHalf;
    const int lb_rel = half * (int)pr_half_bytes(K);
    const int lb = lb_rel + lds_address_of(lds_all); // absolute LDS address of this frame's region
    lds_byte_t* lds = (lds_byte_t*)lds_all + lb_rel; // (address-space-3 typed pointers: ldpc_kernel.hpp, lds_byte_t)
    // Round 5: a sign-vector area per FRAME. With one area per workgroup the two frames took turns through the full syndrome test (three
    // barriers each), and a full test cost both of them 27 us -- three quarters of an update sweep of short 1/4 (measured with a build that
    // runs it after every update): at the operating point, where converged frames pass the pre-test until their group stops, that was a
    // third of the decode (config4_awgn at 0.65 of the proportional rate).
    lds_u32_t* sv_base = reinterpret_cast<lds_u32_t*>((lds_byte_t*)lds_all + 2 * (int)pr_half_bytes(K));
    const bool sv_shared = (stop_on_good & 8) != 0; // uniform: one area for the workgroup (see pr_lds_bytes)
    lds_u32_t* sv = sv_base + (sv_shared ? 0 : half) * (N / kM) * kSvWords;
    volatile lds_i32_t* flags_all = reinterpret_cast<volatile lds_i32_t*>(sv_base + (sv_shared ? 1 : 2) * (N / kM) * kSvWords);
    volatile lds_i32_t* flags = flags_all + 8 * half;        // [0] bad-or, [1] finished, [2] pre-test failed, [3] full test needed
    volatile lds_i32_t* other_flags = flags_all + 8 * (1 - half);
    const int f = 2 * blockIdx.x + half;
    const bool have_frame = f < n_frames;
    const int lane = tid & 63;
    const int NGD = K / kM, NG = N / kM;
    const bool active = tid < kM;
    const int row = tid < kM ? tid : kM - 1; // threads 360..383 mirror row 359 (ldpc_kernel.hpp)
    const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6); // wave of the frame, known to be uniform
    const uint32_t* srec = V2 ? wrecs + (size_t)wave_u * RSW : recs; // the sweep's records: per (layer, wave) with packed nodes, else per layer

    uint32_t* msg_base = msgs + (size_t)(have_frame ? f : 0) * q * RW * kMsgStride;
    int it = 0, tgt = 0;
    int pr_epoch = 0; // this kernel keeps the hardware barrier (check_node_hazard takes a frame-barrier state)
    bool finished = !have_frame;
    if (have_frame) {
        tgt = target ? target[f] : cap;
        if (llr_in) {
            const int8_t* src = llr_in + (size_t)f * N;
            const uint2* src8 = reinterpret_cast<const uint2*>(src);
            for (int c = tid; c < K / 8; c += kHalf) {
                uint2 v = src8[c];
                v.x ^= 0x80808080u; v.y ^= 0x80808080u;
                *reinterpret_cast<lds_v2u_t*>(lds + 8 * c) = (v2u32){ v.x, v.y };
            }
            if (active) {
                // parity[q*j + i] = P[i][j] (layered_decoder.hh:150-152): row q-1 to LDS, row i < q-1 to byte 7 of
                // record i+1; all message bytes start at offset-binary zero (bnl = 0, layered_decoder.hh:27-31)
                const int8_t* pj = src + K + (size_t)q * tid;
                if constexpr (W1) {
                    msg_base[tid] = 0x80820820u; // four zero messages (field value 32), parity byte unused in record 0
                    for (int i = 0; i < q - 1; i++) msg_base[(i + 1) * kMsgStride + tid] = 0x00820820u | ((uint32_t)((uint8_t)pj[i] ^ 0x80u) << 24);
                } else {
                msg_base[0 * kMsgStride + tid] = 0x80808080u;
                msg_base[1 * kMsgStride + tid] = 0x80808080u;
                for (int i = 0; i < q - 1; i++) {
                    const uint32_t b = (uint8_t)pj[i] ^ 0x80u;
                    // zero messages in the format of the (layer, wave) that owns the record: offset binary, or two's complement for a packed node
                    const uint32_t z = (V2 && ((srec[(size_t)(i + 1) * LS] >> 13) & 1u)) ? 0u : 0x80808080u;
                    msg_base[((i + 1) * MW + 0) * kMsgStride + tid] = z;
                    msg_base[((i + 1) * MW + 1) * kMsgStride + tid] = (z & 0x00ffffffu) | (b << 24);
                }
                }
                lds[K + tid] = (uint8_t)pj[q - 1] ^ 0x80u;
            }
        } else {
            it = iters[f];
            if (it >= tgt) finished = true;
            else {
                // resume: information LLRs and parity row q-1 come back from the state buffer; the other parity rows
                // are still in the message records
                const uint2* src8 = reinterpret_cast<const uint2*>(state + (size_t)f * N);
                for (int c = tid; c < K / 8; c += kHalf) { const uint2 v = src8[c]; *reinterpret_cast<lds_v2u_t*>(lds + 8 * c) = (v2u32){ v.x, v.y }; }
                if (active) lds[K + tid] = state[(size_t)f * N + K + kM * (q - 1) + tid];
            }
        }
    }
    const bool untouched = finished;
    if (tid == 0) { flags[0] = 0; flags[2] = 0; flags[3] = 0; flags[1] = finished ? 1 : 0; }
    __syncthreads();

    bool is_good = false;
    for (;;) {
        // ---- syndrome test: pre-test on one layer, full test only for frames that pass it (see ldpc_kernel.hpp) ----
        const bool need_synd = !finished && ((stop_on_good & 1) || it >= tgt);
        if (need_synd && active) {
            const int i0 = it % q;
            const uint32_t* rec = recs + (size_t)i0 * RS;
            const int deg = (int)(rec[0] & 0xffu) + 2;
            uint32_t x = 0, z = 0;
            for (int k = 0; k < deg; k++) {
                uint32_t v;
                if (k == deg - 2 && i0 != q - 1) v = msg_base[((i0 + 1) * RW + PW) * kMsgStride + tid] >> 24;       // own parity P[i0]
                else if (k == deg - 1 && i0 != 0) v = msg_base[(i0 * RW + PW) * kMsgStride + tid] >> 24;           // previous parity P[i0-1]
                else {
                    const int a0 = tid + (int)rec[4 + 2 * k] - ((uint32_t)tid < rec[5 + 2 * k] ? 0 : kM);
                    v = lds[a0];
                    if (i0 == 0 && k == deg - 1 && tid == 0) v = 0x81u; // check (0,0) has no previous parity
                }
                x ^= v;
                z |= (v == 0x80u);
            }
            const int bad_pre = (int)((((x >> 7) ^ (uint32_t)deg) & 1u) | z);
            if (__ballot(bad_pre) != 0 && lane == 0) flags[2] = 1;
        }
        __syncthreads();
#ifdef DVBS2_EXP_ALWAYS_FULL // timing experiment (same results): the full test after every update, whatever the pre-test says
        const bool need_full = need_synd;
#else
        const bool need_full = need_synd && flags[2] == 0;
#endif
        if (tid == 0) flags[3] = need_full ? 1 : 0;
        __syncthreads();
        if (flags[3] != 0 || other_flags[3] != 0) { // uniform over the workgroup
            for (int h = 0; h < (sv_shared ? 2 : 1); h++) { // both frames at once when each has its own sign-vector area, else in turns
                const bool mine = need_full && (!sv_shared || half == h);
                if (mine) {
                    const unsigned long long zero_any = pr_sign_vectors<W1>(lds, sv, msg_base, K, N, q, tid);
                    if (zero_any != 0 && lane == 0) flags[0] = 1;
                }
                __syncthreads();
                if (mine && tid < NG) {
                    lds_u32_t* p = sv + tid * kSvWords;
                    const uint32_t w0 = p[0], w1 = p[1];
                    p[11] = (p[11] & 0xffu) | (w0 << 8);
                    p[12] = (w0 >> 24) | (w1 << 8);
                }
                __syncthreads();
                if (mine) {
                    int bad = 0;
                    for (int item = tid; item < q * 12; item += kHalf) {
                        const int i = item / 12, w = item - 12 * i;
                        const uint32_t* rec = recs + (size_t)i * RS;
                        const int deg = (int)(rec[0] & 0xffu) + 2;
                        uint32_t acc = 0;
                        for (int k = 0; k < deg; k++) {
                            int g, rot;
                            if (k == deg - 2) { g = NGD + i; rot = 0; }                                   // own parity row i
                            else if (k == deg - 1) { g = NGD + (i ? i - 1 : q - 1); rot = i ? 0 : 359; } // previous parity
                            else { rot = kM - (int)rec[5 + 2 * k]; g = ((int)rec[4 + 2 * k] - rot) / kM; }
                            const int t0 = wrap360(32 * w + rot);
                            const lds_u32_t* p = sv + g * kSvWords + (t0 >> 5);
                            uint32_t x = __funnelshift_r(p[0], p[1], t0 & 31);
                            if (i == 0 && k == deg - 1 && w == 0) x &= ~1u;
                            acc ^= x;
                        }
                        if (w == 11) acc &= 0xffu;
                        bad |= acc != 0;
                    }
                    if (__ballot(bad) != 0 && lane == 0) flags[0] = 1;
                }
                __syncthreads();
            }
        }
        if (need_synd) is_good = need_full && flags[0] == 0;
        const bool gs = (stop_on_good & 5) == 5;
        if (!finished && (it >= tgt || (!gs && (stop_on_good & 1) && is_good))) finished = true;
        __syncthreads();
        if (tid < 64) { // the frame's first wave (group_decide reads one status word per lane)
            int fin = finished ? 1 : 0;
            if (gs && !finished) {
                const uint32_t* hd = recs - kRecHeaderWords;
                const int* iters0 = reinterpret_cast<const int*>(((unsigned long long)hd[1] << 32) | hd[0]);
                int* status = reinterpret_cast<int*>(((unsigned long long)hd[3] << 32) | hd[2]);
                const int G = (int)hd[4];
                const int g = f / G;
                fin = group_decide(status + (iters - iters0) + g * G, min(G, n_frames - g * G), f - g * G, it, is_good, tid, (int)hd[5]) != 0;
            }
            if (tid == 0) { flags[0] = 0; flags[2] = 0; flags[1] = fin; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // LDS only: the group report above is not waited for (ldpc_kernel.hpp, frame_barrier_lds)
        if (gs) finished = flags[1] != 0;
        if (finished && other_flags[1]) break;

        // ---- one update sweep ----
        finished = __builtin_amdgcn_readfirstlane((int)finished) != 0; // (uniform over the frame; read from LDS above)
        const uint32_t roff = (uint32_t)row * 4u;
        const bool work = !finished;
        uint32_t pre1[RW], pre2[RW]; // records of the next two layers for this row
        int carry = 0x80;
        if (work) {
#pragma unroll
            for (int w = 0; w < RW; w++) { pre1[w] = msg_base[w * kMsgStride + row]; pre2[w] = msg_base[(RW + w) * kMsgStride + row]; }
        }
        uint32_t nhdr = srec[0];
        uint32_t nent[2 * DMAX];
#pragma unroll
        for (int k = 0; k < 2 * DMAX; k++) nent[k] = srec[4 + k];
        // The record a layer produces is STORED at the head of the next layer, behind an explicit s_waitcnt vmcnt(0) (see
        // DVBS2_WAIT_BEFORE_STORE in ldpc_kernel.hpp: the compiler waits for the prefetched record with vmcnt(0) at the head of a layer,
        // i.e. right behind the store the previous layer has just issued; stored here instead, everything that wait covers is a layer old).
        // Here the wait cannot simply go in front of the store at the END of the layer: the layers of these tables are short and the
        // prefetch issued at their head would not be back yet.
        constexpr bool kDeferStore = DVBS2_PR_DEFER_STORE != 0;
        uint32_t pend[RW];
        int pend_i = 0;
        bool pend_on = false; // uniform
#pragma unroll
        for (int w = 0; w < RW; w++) pend[w] = 0;
        for (int i = 0; i < q; i++) {
            const uint32_t hdr = nhdr;
            uint32_t ent[2 * DMAX];
#pragma unroll
            for (int k = 0; k < 2 * DMAX; k++) ent[k] = nent[k];
            {
                const uint32_t* nrec = srec + (size_t)(i + 1 < q ? i + 1 : 0) * LS;
                nhdr = nrec[0];
#pragma unroll
                for (int k = 0; k < 2 * DMAX; k++) nent[k] = nrec[4 + k];
            }
            const int deg = (int)(hdr & 0xffu) + 2;
            const int nc = (int)((hdr >> 8) & 0xfu);
            const int block = (int)(hdr >> 16);
            const bool first_layer = (i == 0), last_layer = (i == q - 1);
            uint32_t* mp = msg_base + (size_t)i * RW * kMsgStride;
            if (hdr & 0x8000u) __syncthreads();
            const int jj = row;
            uint32_t mw[MW], nm[MW];
            uint32_t x6 = 0;
            if constexpr (W1) {
                x6 = pre1[0];
                mw[0] = 0x80808080u; mw[1] = 0x80808080u; // (hazard layers expand the record below)
                pre1[0] = pre2[0];
            } else {
#pragma unroll
                for (int w = 0; w < MW; w++) { mw[w] = work ? pre1[w] : 0x80808080u; pre1[w] = pre2[w]; }
            }
            const int own_in = (int)(pre1[PW] >> 24); // top byte of record i+1 = P[i] (not used by the last layer)
            if constexpr (kDeferStore) {
                asm volatile("" ::: "memory"); __builtin_amdgcn_s_waitcnt(0x0f70); asm volatile("" ::: "memory"); // vmcnt(0), on every path
                if (pend_on) { // (pend_on: work && i > 0, the record of layer i - 1)
#pragma unroll
                    for (int w = 0; w < RW; w++) pr_st(mp - (RW - w) * kMsgStride, roff, pend[w]);
                }
            }
            if (work && i + 2 < q) {
#pragma unroll
                for (int w = 0; w < RW; w++) pre2[w] = pr_ld(mp + (2 * RW + w) * kMsgStride, roff);
            }
            uint32_t y6 = 0;
            bool have_y6 = false;
            if (block >= kM) {
                if constexpr (W1) {
                    if (work) { // (the record still sits in pre-expansion form in x6)
                        have_y6 = true;
                        if (deg == 4) { if (first_layer) y6 = check_node_pr6<4, true, false>(lds_all, ent, jj, lb, x6, own_in, &carry); else if (last_layer) y6 = check_node_pr6<4, false, true>(lds_all, ent, jj, lb, x6, own_in, &carry); else y6 = check_node_pr6<4, false, false>(lds_all, ent, jj, lb, x6, own_in, &carry); }
                        else { if (first_layer) y6 = check_node_pr6<3, true, false>(lds_all, ent, jj, lb, x6, own_in, &carry); else if (last_layer) y6 = check_node_pr6<3, false, true>(lds_all, ent, jj, lb, x6, own_in, &carry); else y6 = check_node_pr6<3, false, false>(lds_all, ent, jj, lb, x6, own_in, &carry); }
                    }
                } else {
                    if (work) {
                        if (V2 && ((hdr >> 13) & 1u)) { // this wave's record is in the packed node's format (a regular layer that is neither the first nor the last)
                            switch (deg) {
                                case 5: check_node_v2_pr<5>(ent, jj + lb, mw, nm, own_in, &carry); break;
                                case 6: check_node_v2_pr<6>(ent, jj + lb, mw, nm, own_in, &carry); break;
                                case 7: check_node_v2_pr<7>(ent, jj + lb, mw, nm, own_in, &carry); break;
                                default: break;
                            }
                        } else { DVBS2_PR_SWITCH }
                    }
                }
            } else {
                if constexpr (W1) { if (work) pr_w1_expand(x6, mw); }
                DVBS2_PRH_SWITCH
            }
            if (work) {
                if constexpr (kDeferStore) {
                    if constexpr (W1) pend[0] = have_y6 ? y6 : pr_w1_compress(nm);
                    else {
#pragma unroll
                        for (int w = 0; w < MW; w++) pend[w] = nm[w];
                    }
                    pend_i = i; pend_on = true;
                } else
                if constexpr (W1) mp[jj] = have_y6 ? y6 : pr_w1_compress(nm);
                else {
#pragma unroll
                    for (int w = 0; w < MW; w++) mp[w * kMsgStride + jj] = nm[w];
                }
            }
        }
        if constexpr (kDeferStore) {
            if (pend_on) { // the last layer's record
                uint32_t* pp = msg_base + (size_t)pend_i * RW * kMsgStride;
#pragma unroll
                for (int w = 0; w < RW; w++) pp[w * kMsgStride + row] = pend[w];
            }
        }
        __syncthreads();
        if (!finished) it++;
    }

    if (have_frame && !untouched) {
        if (tid == 0) { iters[f] = it; good[f] = is_good ? 1 : 0; }
        uint8_t* dst = state + (size_t)f * N;
        for (int c = tid; c < K / 8; c += kHalf) { const v2u32 v = *reinterpret_cast<const lds_v2u_t*>(lds + 8 * c); reinterpret_cast<uint2*>(dst)[c] = make_uint2(v.x, v.y); }
        if (active) {
            for (int i = 0; i < q - 1; i++) dst[K + kM * i + tid] = (uint8_t)(msg_base[((i + 1) * RW + PW) * kMsgStride + tid] >> 24);
            dst[K + kM * (q - 1) + tid] = lds[K + tid];
        }
    }
}

#endif // DVBS2_LDPC_INSTANTIATE_PR

hipError_t ldpc_pr_prepare(size_t lds_bytes);
void ldpc_pr_launch(const LdpcLaunch& a); // a.v2 = one-dword records (check degree <= 4)

#ifdef DVBS2_LDPC_INSTANTIATE_PR
hipError_t ldpc_pr_prepare(size_t lds_bytes)
{
    hipError_t e = hipFuncSetAttribute((const void*)ldpc_layered_pr_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ldpc_layered_pr_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)ldpc_layered_pr_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e == hipSuccess && getenv("DVBS2_OCC")) {
        int nb = -1;
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ldpc_layered_pr_kernel<false>, kThreads, lds_bytes);
        fprintf(stderr, "[pr kernel] lds %zu bytes, occupancy API: %d workgroups per CU\n", lds_bytes, nb);
    }
    return e;
}
void ldpc_pr_launch(const LdpcLaunch& a)
{
    if (a.v2) hipLaunchKernelGGL(ldpc_layered_pr_kernel<true>, dim3((a.n_frames + 1) / 2), dim3(kThreads), a.lds_bytes, a.stream, a.recs, a.wrecs, a.llr_in, a.state, a.msgs,
                                 a.iters, a.good, a.target, a.n_frames, a.N, a.K, a.q, a.cap, a.stop_on_good);
    else if (a.pr_packed)
        hipLaunchKernelGGL((ldpc_layered_pr_kernel<false, true>), dim3((a.n_frames + 1) / 2), dim3(kThreads), a.lds_bytes, a.stream, a.recs, a.wrecs, a.llr_in, a.state, a.msgs,
                           a.iters, a.good, a.target, a.n_frames, a.N, a.K, a.q, a.cap, a.stop_on_good);
    else hipLaunchKernelGGL(ldpc_layered_pr_kernel<false>, dim3((a.n_frames + 1) / 2), dim3(kThreads), a.lds_bytes, a.stream, a.recs, a.wrecs, a.llr_in, a.state, a.msgs,
                            a.iters, a.good, a.target, a.n_frames, a.N, a.K, a.q, a.cap, a.stop_on_good);
}
#endif

} // namespace dvbs2
