// bch_hip.hip -- GF(2^m) BCH syndrome / Berlekamp / Chien decoder for the DVB-S2/S2X outer code on gfx950.
//
// Bit-exact contract: bch_codec<uint32_t, bitset256_t>::decode(u8_cptr_t, u8_ptr_t) of the reference
// (lib/bch.cc:468-487) including its behaviour beyond t errors (partial corrections, return -1) and the
// two places where it throws (reported as -2 instead of unwinding through the C ABI).
//
// Mapping: persistent workgroups of 1024 threads, one per CU, each keeping the antilog table of the field
// (2^m - 1 entries, 128 KB for GF(2^16)) in LDS for its whole life; frames are dealt round-robin.
//   syndromes   batches of >= 32 frames: all odd syndromes of all frames as ONE binary matrix product on the matrix cores
//               (bch_syndrome_kernel, below); smaller batches inside the per-frame kernel:
//               S_i = r(alpha^i), i odd: every thread owns a stride of codeword bytes and adds
//               alpha^(i*e mod P) for each set bit of exponent e (== remainder-then-evaluate of the
//               reference, lib/bch.cc:176-189,217-222: rem(alpha^i) = r(alpha^i), and rem == 0 <=> all S_i == 0);
//               S_2i = S_i^2.
//   sigma(x)    simplified Berlekamp table (lib/bch.cc:225-304), one wavefront (lane = coefficient column), log/antilog arithmetic.
//   roots       degree 1 and 2 closed forms (lib/bch.cc:316-367); otherwise a Chien search over the
//               exponents [s+1, n+s] (lib/bch.cc:376-384, lib/gf.cc:376-401), all threads, LDS gathers.
//   correction  message bits only, network bit order (lib/bch.cc:429-452).
#include "bch_hip.h"
#include "device_guard.h"
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace dvbs2 {

bool BchCode::build(int m_, uint32_t prim_poly, int t_, int n_, std::string* err)
{
    m = m_; t = t_;
    if (m < 3 || m > 16 || t < 1 || t > 12) { *err = "unsupported GF(2^m) dimension or t"; return false; }
    P = (1 << m) - 1;
    antilog.assign(P, 0); log.assign(P + 1, 0);
    const uint32_t low = prim_poly ^ (1u << m);
    uint32_t x = 1;
    for (int i = 0; i < P; i++) {
        antilog[i] = (uint16_t)x; log[x] = (uint16_t)i;
        x = ((x << 1) & (uint32_t)P) ^ ((x >> (m - 1)) * low);
    }
    auto gmul = [&](uint32_t a, uint32_t b) -> uint32_t { return (!a || !b) ? 0u : antilog[(log[a] + log[b]) % P]; };
    // generator polynomial
    std::vector<uint8_t> seen(P + 1, 0);
    gen.assign(1, 1);
    for (int i = 0; i < t; i++) {
        uint32_t e = (uint32_t)((2 * i + 1) % P);
        if (seen[antilog[e]]) continue;
        std::vector<uint32_t> conj;
        uint32_t ee = e;
        for (int j = 0; j < m; j++) {
            uint32_t el = antilog[ee];
            if (std::find(conj.begin(), conj.end(), el) != conj.end()) break;
            conj.push_back(el); seen[el] = 1;
            ee = (uint32_t)(((uint64_t)ee * 2) % P);
        }
        std::vector<uint32_t> mp(1, 1);
        for (uint32_t c : conj) {
            std::vector<uint32_t> nx(mp.size() + 1, 0);
            for (size_t d = 0; d < mp.size(); d++) { nx[d + 1] ^= mp[d]; nx[d] ^= gmul(mp[d], c); }
            mp.swap(nx);
        }
        std::vector<uint8_t> ng(gen.size() + mp.size() - 1, 0);
        for (size_t a = 0; a < gen.size(); a++) if (gen[a]) for (size_t d = 0; d < mp.size(); d++) {
            if (mp[d] > 1) { *err = "minimal polynomial is not binary"; return false; }
            ng[a + d] ^= (uint8_t)mp[d];
        }
        gen.swap(ng);
    }
    gdeg = (int)gen.size() - 1;
    n = n_ ? n_ : P;
    if (n > P) { *err = "Codeword length n exceeds the maximum of (2^m - 1)"; return false; }
    if (n <= gdeg) { *err = "Codeword length n must be greater than the generator polynomial's degree"; return false; }
    s = P - n; k = n - gdeg;
    quad.assign(P + 1, 0);
    for (uint32_t r = 0; r <= (uint32_t)P; r++) quad[gmul(r, r) ^ r] = (uint16_t)r;
    return true;
}

constexpr int kBchThreads = 1024;
constexpr int kMaxT = 12;
constexpr int kBchWorkWords = 640; // dwords of per-workgroup scratch between the antilog table and the codeword bytes

struct BchArgs {
    const uint16_t* antilog; const uint16_t* log; const uint16_t* quad;
    const uint8_t* cw; uint8_t* msg; int32_t* corr;
    const uint8_t* llr_state; int llr_stride; // cw == nullptr: the codeword bits are the hard decisions of these offset-binary LLR
                                              // bytes (the LDPC decoder's state, information part in natural order): bit = byte < 0x80
    const uint4* hcol;         // parity-check columns: hcol[2 e], hcol[2 e + 1] = alpha^(e), alpha^(3 e), .., alpha^((2t-1) e) as 16-bit halves (32 B per bit)
    const uint8_t* descramble; // k/8 bytes of the BB PRBS or nullptr (fused bbdescrambler_bb)
    const uint32_t* synd;      // non-null: the odd syndromes of every frame, computed by bch_syndrome_kernel (8 dwords per frame: S_(2u+1)
                               // in half u & 1 of dword u >> 1)
    int n_frames, m, P, t, n, k, s;
};

__device__ __forceinline__ uint32_t modP(uint32_t x, int m, uint32_t P)
{
    x = (x & P) + (x >> m);
    x = (x & P) + (x >> m);
    return x >= P ? x - P : x;
}

// xor / max over the 64 lanes of a wavefront, result in every lane: four DPP exchange stages inside the rows of 16 lanes, then the
// four row results through v_readlane
template <int CTRL> __device__ __forceinline__ int bch_dpp(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ uint32_t wave_xor(uint32_t x)
{
    int v = (int)x;
    v ^= bch_dpp<0xB1>(v); v ^= bch_dpp<0x4E>(v); v ^= bch_dpp<0x141>(v); v ^= bch_dpp<0x140>(v); // quad xor 1, xor 2, half-row mirror, row mirror
    return (uint32_t)(__builtin_amdgcn_readlane(v, 0) ^ __builtin_amdgcn_readlane(v, 16) ^ __builtin_amdgcn_readlane(v, 32) ^ __builtin_amdgcn_readlane(v, 48));
}
__device__ __forceinline__ int wave_max(int x)
{
    int v = x;
    v = max(v, bch_dpp<0xB1>(v)); v = max(v, bch_dpp<0x4E>(v)); v = max(v, bch_dpp<0x141>(v)); v = max(v, bch_dpp<0x140>(v));
    return max(max(__builtin_amdgcn_readlane(v, 0), __builtin_amdgcn_readlane(v, 16)), max(__builtin_amdgcn_readlane(v, 32), __builtin_amdgcn_readlane(v, 48)));
}

// ---- odd syndromes of a whole batch as ONE matrix product over GF(2) on the matrix cores ----
// S = H c: H is the 16 t x n binary parity-check matrix (row 16 u + b = bit b of alpha^((2u+1) e) along the codeword), c the n x n_frames
// matrix of hard decisions. One 32-byte table read per SET BIT and frame (above) costs a clean 8PSK 3/4 batch 0.55 ms per 4096 frames; as
// a product the table is read once per 32 frames: v_mfma_i32_32x32x32_i8 on 0 / 1 bytes, the syndrome bit is the PARITY of the exact
// integer sum (<= n < 2^31). A = 32 rows of H (bytes, zero-padded to a multiple of 128 columns and 32 rows), B = 32 frames: the LLR
// bytes of the LDPC decoder's state turn into 0 / 1 with two instructions per dword (or packed codeword bytes are spread); both operands
// give lane (r, h) the 64 consecutive columns k0 + 64 h .. of row / frame r, sixteen per instruction -- the K order inside an
// instruction does not matter as long as A and B agree. A wave owns 32 frames x all rows x a chunk of the columns and xors its parities
// into the frame's eight syndrome dwords (zeroed by the host before the launch); the four waves of a workgroup share the rows of H.
using bch_v4i = __attribute__((ext_vector_type(4))) int;
using bch_v16i = __attribute__((ext_vector_type(16))) int;
constexpr int kSynK = 128; // columns per step of a wave

template <int RT /*row tiles of 32: 16 t / 32 rounded up*/, bool PACKED>
__global__ __launch_bounds__(256, 2) void bch_syndrome_kernel(const int8_t* __restrict__ H, int Kp /*padded columns*/, const uint8_t* __restrict__ src,
                                                              size_t stride /*bytes per frame of src*/, int nb /*PACKED: codeword bytes*/,
                                                              int n_frames, int steps_per_chunk, uint32_t* __restrict__ synd)
{
    // The four waves of a workgroup own four tiles of 32 frames and the SAME columns: the 32 RT x 128 bytes of H that a step needs are
    // fetched once per workgroup -- wave w brings sixteen-column slice w of every row tile -- and handed over in LDS already in operand
    // order (two buffers, one barrier per step). Straight from the L2 every wave read them itself: 4.7 KB through the vector cache per
    // instruction, 0.15 ms per 4096 frames of 8PSK 3/4.
    __shared__ bch_v4i hs[2][RT * 4 * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ft = (int)blockIdx.x * 4 + wave;
    const bool wave_on = ft * 32 < n_frames; // (a wave without frames still fetches its share of H and takes part in the barriers)
    const int r = lane & 31, h = lane >> 5;
    int f = ft * 32 + r;
    const bool fvalid = f < n_frames;
    if (!fvalid) f = n_frames - 1; // (reads a valid frame, contributes nothing)
    bch_v16i acc[RT];
#pragma unroll
    for (int rt = 0; rt < RT; rt++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[rt][i] = 0;
    const int s0 = (int)blockIdx.y * steps_per_chunk, s1 = min(s0 + steps_per_chunk, Kp / kSynK);
    const uint8_t* fsrc = src + (size_t)f * stride;
    constexpr int NRAW = PACKED ? 8 : 16;
    bch_v4i hreg[RT];
    uint32_t raw[NRAW];
    auto fetch = [&](int st) {
        const int k = st * kSynK + 64 * h;
#pragma unroll
        for (int rt = 0; rt < RT; rt++) hreg[rt] = *reinterpret_cast<const bch_v4i*>(H + (size_t)(rt * 32 + r) * (size_t)Kp + k + 16 * wave);
        if (wave_on) {
            if constexpr (!PACKED) {
                const uint2* p = reinterpret_cast<const uint2*>(fsrc + k);
#pragma unroll
                for (int i = 0; i < 8; i++) { const uint2 w = p[i]; raw[2 * i] = w.x; raw[2 * i + 1] = w.y; }
            } else {
#pragma unroll
                for (int i = 0; i < 8; i++) { const int bi = (k >> 3) + i; raw[i] = bi < nb ? (uint32_t)fsrc[bi] : 0u; }
            }
        }
    };
    auto publish = [&](int buf) {
#pragma unroll
        for (int rt = 0; rt < RT; rt++) hs[buf][(rt * 4 + wave) * 64 + lane] = hreg[rt];
    };
    int buf = 0;
    if (s0 < s1) { fetch(s0); publish(0); }
    __syncthreads();
    for (int st = s0; st < s1; st++) {
        uint32_t b[16];
        if constexpr (!PACKED) { // offset-binary LLR bytes (8-byte aligned rows): bit = 1 where the LLR is negative = bit 7 clear
#pragma unroll
            for (int i = 0; i < 16; i++) b[i] = (~raw[i] >> 7) & 0x01010101u;
        } else { // packed bytes, first bit of the stream = bit 7: nibble n -> bytes (n >> 3) & 1, (n >> 2) & 1, (n >> 1) & 1, n & 1
            auto spread = [](uint32_t nib) { return ((nib >> 3) & 1u) | ((nib & 4u) << 6) | ((nib & 2u) << 15) | ((nib & 1u) << 24); };
#pragma unroll
            for (int i = 0; i < 8; i++) { b[2 * i] = spread(raw[i] >> 4); b[2 * i + 1] = spread(raw[i] & 15u); }
        }
        if (st + 1 < s1) fetch(st + 1); // in flight while this step multiplies
        if (wave_on) {
#pragma unroll
            for (int rt = 0; rt < RT; rt++) {
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    bch_v4i bv; bv[0] = (int)b[4 * i]; bv[1] = (int)b[4 * i + 1]; bv[2] = (int)b[4 * i + 2]; bv[3] = (int)b[4 * i + 3];
                    acc[rt] = __builtin_amdgcn_mfma_i32_32x32x32_i8(hs[buf][(rt * 4 + i) * 64 + lane], bv, acc[rt], 0, 0, 0);
                }
            }
        }
        if (st + 1 < s1) publish(buf ^ 1); // (everybody left that buffer at the previous barrier)
        __syncthreads();
        buf ^= 1;
    }
    // D: column (frame) = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
    if (fvalid && wave_on) {
#pragma unroll
        for (int rt = 0; rt < RT; rt++) {
            uint32_t bits = 0;
#pragma unroll
            for (int i = 0; i < 16; i++) bits |= ((uint32_t)acc[rt][i] & 1u) << ((i & 3) + 8 * (i >> 2));
            bits <<= 4 * h;
            if (bits) atomicXor(&synd[(size_t)f * 8 + rt], bits);
        }
    }
}

__global__ __launch_bounds__(kBchThreads) void bch_decode_kernel(BchArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
    const int tid = threadIdx.x;
    const uint32_t P = (uint32_t)a.P;
    const int m = a.m, t = a.t, n = a.n, k = a.k;
    const int nb = n / 8, kb = k / 8;
    uint16_t* al = reinterpret_cast<uint16_t*>(smem);                  // antilog, P entries
    uint32_t* w = reinterpret_cast<uint32_t*>(smem + (((size_t)P * 2 + 15) & ~(size_t)15));
    uint32_t* S = w;            // [2t] syndromes S_1..S_2t
    uint32_t* lsig = w + 32;    // [t+1] log of sigma coefficients, 0xffffffff for zero
    uint32_t* roots = w + 64;   // [<=16] root exponents found by the Chien search
    int* ctl = reinterpret_cast<int*>(w + 96); // [0] degree, [1] nroots, [2] mode, [3] status
    // Berlekamp table rows at their FULL width: row mu + 1 can reach degree 2 mu + 1 (S_1 .. S_2mu = 0, S_2mu+1 != 0), so the
    // last row can reach 2t - 1 > t; the reference keeps every coefficient (gf2m_poly, lib/bch.cc:286-297) and "degree > t"
    // (lib/bch.cc:313-314) must be decided on the true degree
    constexpr int kSigW = 2 * kMaxT + 4;
    // (w + 104 .. w + 159: the per-row bookkeeping of the one-thread Berlekamp of round 1; it lives in registers of lane `row` now)
    uint32_t (*sg)[kSigW] = reinterpret_cast<uint32_t (*)[kSigW]>(w + 160); // [kMaxT + 3][kSigW] = 420 dwords
    uint8_t* cwl = reinterpret_cast<uint8_t*>(w + kBchWorkWords);       // codeword bytes

    for (uint32_t i = tid; i < P; i += kBchThreads) al[i] = a.antilog[i];
    if (tid == 0) al[P] = 0; // (the two bytes of padding behind the table) the entry absent Chien terms read

    for (int f = blockIdx.x; f < a.n_frames; f += gridDim.x) {
        __syncthreads();
        uint8_t* out = a.msg + (size_t)f * kb;
        for (int b = tid; b < nb; b += kBchThreads) { // lib/bch.cc:471 (+ lib/bbdescrambler_bb_impl.cc:74-78 when fused)
            uint8_t v;
            if (a.cw) v = a.cw[(size_t)f * nb + b];
            else { // hard decision + MSB-first packing of ldpc_decoder_bb (lib/ldpc_decoder_bb_impl.cc:432-442), fused here
                const uint2 w = *reinterpret_cast<const uint2*>(a.llr_state + (size_t)f * a.llr_stride + 8 * b);
                const uint32_t lo = ~w.x & 0x80808080u, hi = ~w.y & 0x80808080u; // bit 7 of every byte: 1 where the LLR is negative
                v = (uint8_t)((((lo >> 7) & 1u) << 7) | (((lo >> 15) & 1u) << 6) | (((lo >> 23) & 1u) << 5) | ((lo >> 31) << 4) |
                              (((hi >> 7) & 1u) << 3) | (((hi >> 15) & 1u) << 2) | (((hi >> 23) & 1u) << 1) | (hi >> 31));
            }
            cwl[b] = v;
            if (b < kb) out[b] = a.descramble ? (uint8_t)(v ^ a.descramble[b]) : v;
        }
        if (tid < 2 * kMaxT) S[tid] = 0;
        if (tid == 0) { ctl[0] = 0; ctl[1] = 0; ctl[2] = 0; ctl[3] = 0; }
        __syncthreads();

        // ---- odd syndromes ----
        // S_(2u+1) = sum over the set bits (exponent e) of alpha^((2u+1) e): the xor of the parity-check matrix columns of the set bits.
        // Round 5: the columns come from a table in global memory (32 bytes per bit: t 16-bit field elements; 1.2-1.9 MB per code, resident in
        // the L2) -- one 32-byte read per set bit instead of t exponent products, reductions mod P and random 16-bit LDS gathers: a real
        // codeword has n / 2 set bits, and the gathers made the syndromes 0.9 ms per 4096 frames of 8PSK 3/4 (a ninth of the chain's step at
        // its operating point) whether the word was clean or not. (Measured 0.11 ms "when clean" in earlier rounds was the all-zero word.)
        if (a.synd) { // computed for the whole batch by bch_syndrome_kernel
            if (tid < t) S[2 * tid] = (a.synd[(size_t)f * 8 + (tid >> 1)] >> (16 * (tid & 1))) & 0xffffu;
        } else {
        uint32_t acc[kMaxT / 2 + 2];
#pragma unroll
        for (int u = 0; u < kMaxT / 2 + 2; u++) acc[u] = 0;
        for (int b = tid; b < nb; b += kBchThreads) {
            uint32_t v = cwl[b];
            while (v) {
                const int hb = 31 - __clz((int)v); // bit value 1<<hb of the byte = stream position 8b + 7 - hb
                v &= ~(1u << hb);
                const uint32_t e = (uint32_t)(n - 1 - (8 * b + 7 - hb));
                const uint4 c0 = a.hcol[2 * e];
                acc[0] ^= c0.x; acc[1] ^= c0.y; acc[2] ^= c0.z; acc[3] ^= c0.w;
                if (t > 8) { const uint4 c1 = a.hcol[2 * e + 1]; acc[4] ^= c1.x; acc[5] ^= c1.y; acc[6] ^= c1.z; acc[7] ^= c1.w; }
            }
        }
#pragma unroll
        for (int w2 = 0; w2 < kMaxT / 2; w2++) {
            if (2 * w2 < t) {
                uint32_t v = acc[w2];
                for (int off = 32; off; off >>= 1) v ^= __shfl_xor((int)v, off);
                if ((tid & 63) == 0) { // S_(2u+1) lives at S[2u]; word w2 holds u = 2 w2 (low half) and 2 w2 + 1 (high half)
                    if (v & 0xffffu) atomicXor(&S[4 * w2], v & 0xffffu);
                    if (2 * w2 + 1 < t && (v >> 16)) atomicXor(&S[4 * w2 + 2], v >> 16);
                }
            }
        }
        } // (!a.synd)
        __syncthreads();

        if (tid < 64) { // the first wavefront; lane c owns coefficient column c of the Berlekamp rows and the bookkeeping of row c
            const int lane = tid;
            // Lanes exchange S[] and sg[][] through LDS without a workgroup barrier (one wavefront, LDS operations of a wave
            // execute in program order): the compiler must not move a lane's load across another lane's earlier store, so every
            // hand-over point is a wavefront-scope release / barrier / acquire (no instruction is emitted for the barrier itself).
            auto wave_lds_sync = [] {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            };
            auto lg = [&](uint32_t x) -> uint32_t { return a.log[x]; };
            auto mul = [&](uint32_t x, uint32_t y) -> uint32_t { return (!x || !y) ? 0u : (uint32_t)al[modP(lg(x) + lg(y), m, P)]; };
            auto inv = [&](uint32_t x) -> uint32_t { return (uint32_t)al[modP(P - lg(x), m, P)]; }; // x != 0
            const bool any = __ballot(lane < t && S[2 * (lane < t ? lane : 0)] != 0) != 0;
            if (!any) { if (lane == 0) { ctl[2] = 0; ctl[3] = 0; } } // error-free fast path (lib/bch.cc:181-182,484-486)
            else {
                // even syndromes S_2i = S_i^2: lane i squares S_i; S_i of an even i is itself a square of an earlier pass (chains
                // 1 -> 2 -> 4 -> 8 -> 16 and 3 -> 6 -> 12 -> 24: four passes)
                for (int pass = 0; pass < 4; pass++) {
                    const uint32_t src = (lane >= 1 && lane <= t) ? S[lane - 1] : 0u;
                    const uint32_t sq = mul(src, src);
                    if (lane >= 1 && lane <= t) S[2 * lane - 1] = sq;
                    wave_lds_sync(); // the next pass squares what other lanes wrote in this one
                }
                // ---- simplified Berlekamp (lib/bch.cc:225-304), one wavefront: the discrepancy is an xor over lanes, the choice of
                // rho a max over lanes of (2 rho - degree, rho) -- the reference scans rho downwards and takes a strictly larger
                // difference, i.e. the LARGEST rho among the best --, the row update one multiply per lane ----
                for (int r = 0; r < t + 3; r++) if (lane < kSigW) sg[r][lane] = 0;
                // per-lane bookkeeping of row `lane`: d, degree, 2 mu
                uint32_t d_l = lane == 0 ? 1u : lane == 1 ? S[0] : 0u;
                int dg_l = lane == 2 ? (S[0] ? 1 : 0) : 0;
                const int two_mu_l = lane == 0 ? -1 : 2 * (lane - 1);
                wave_lds_sync(); // (lane 0 overwrites columns other lanes just cleared)
                if (lane == 0) { sg[0][0] = 1; sg[1][0] = 1; sg[2][0] = 1; sg[2][1] = S[0]; }
                wave_lds_sync();
                int dg_row = S[0] ? 1 : 0; // degree of the current row (uniform)
                int row = 2;
                while (row <= t) {
                    const int tm = 2 * (row - 1);
                    const uint32_t mine = lane < kSigW ? sg[row][lane] : 0u;
                    uint32_t term = 0;
                    if (lane == 0) term = S[tm];
                    else if (lane <= dg_row && mine) term = mul(mine, S[tm - lane]);
                    const uint32_t dr = wave_xor(term);
                    if (lane == row) d_l = dr;
                    int dg_next;
                    if (dr == 0) { if (lane < kSigW) sg[row + 1][lane] = mine; dg_next = dg_row; }
                    else {
                        const int key = (lane < row && d_l != 0) ? (((two_mu_l - dg_l + 32) << 8) | lane) : -1;
                        const int best = wave_max(key);
                        const int row_rho = best & 0xff;
                        const int shift = tm - (row_rho == 0 ? -1 : 2 * (row_rho - 1));
                        const uint32_t d_rho = (uint32_t)__builtin_amdgcn_readlane((int)d_l, row_rho);
                        const int dg_rho = __builtin_amdgcn_readlane(dg_l, row_rho);
                        const uint32_t coef = mul(dr, inv(d_rho));
                        uint32_t nv = mine;
                        const int src = lane - shift;
                        if (lane < kSigW && src >= 0 && src <= dg_rho) nv ^= mul(coef, sg[row_rho][src]); // src + shift <= 2t - 1
                        if (lane < kSigW) sg[row + 1][lane] = nv;
                        const unsigned long long nz = __ballot(lane < kSigW && nv != 0);
                        dg_next = nz ? 63 - __clzll((long long)nz) : -1;
                    }
                    if (lane == row + 1) dg_l = dg_next;
                    dg_row = dg_next;
                    row++;
                    wave_lds_sync(); // row `row` is complete before any lane reads it (its own column, or sg[row_rho][src] later)
                }
                if (lane == 0) {
                const int deg = dg_row;
                const uint32_t* sigma = sg[row];
                ctl[0] = deg;
                // ---- error-location numbers (lib/bch.cc:307-385) ----
                int mode = 1, nnum = 0, status = 0; // mode 1: numbers ready in roots[] as bit indices; 2: Chien needed
                uint32_t num[2] = { 0, 0 };
                if (deg > t) { nnum = 0; }
                else if (deg == 1) { num[0] = mul(sigma[1], inv(sigma[0])); nnum = 1; }
                else if (deg == 2) {
                    if (sigma[1] == 0 || sigma[0] == 0) nnum = 0;
                    else {
                        const uint32_t b_over_a = mul(sigma[1], inv(sigma[2]));
                        const uint32_t rr = mul(mul(sigma[0], sigma[2]), inv(mul(sigma[1], sigma[1])));
                        const uint32_t r = a.quad[rr];
                        const uint32_t x0 = mul(r, b_over_a), x1 = mul(b_over_a, r ^ 1u);
                        if (x0 == 0 || x1 == 0) status = -2; // galois_field::inverse(0) throws (lib/gf.h:110)
                        else { num[0] = inv(x0); num[1] = inv(x1); nnum = 2; }
                    }
                } else mode = 2; // (the logarithms of sigma's coefficients: below, one lane each)
                if (mode == 1 && status == 0) {
                    for (int i = 0; i < nnum; i++) roots[i] = lg(num[i]); // bit index = exponent of the number
                    ctl[1] = nnum;
                }
                ctl[2] = mode; ctl[3] = status;
                if (mode == 1 && status == -2) ctl[2] = 3; // nothing to apply
                }
                wave_lds_sync();
                // Chien search ahead: log sigma_j, lane j (thirteen table reads in global memory at once instead of one after the other)
                if (ctl[2] == 2 && lane <= dg_row) { const uint32_t c = sg[row][lane]; lsig[lane] = c ? lg(c) : 0xffffffffu; }
            }
        }
        __syncthreads();
        const int mode = ctl[2], deg = ctl[0];
        if (mode == 0) { if (tid == 0) a.corr[f] = 0; continue; }
        if (mode == 3) { if (tid == 0) a.corr[f] = -2; continue; }
        if (mode == 2) {
            // ---- Chien search over exponents [s+1, n+s]; a degree-deg polynomial has at most deg roots, so
            // collecting all of them equals the reference's early-stopping scan ----
            // sigma(alpha^i) = xor_j alpha^(log sigma_j + i j): a thread walks i = i0, i0 + 1024, ... and keeps the exponent of every
            // term, advanced by (1024 j mod P) per step (an add and a conditional subtract instead of a multiply and a reduction)
            // Round 5: exponents kept DOUBLED (= byte offsets into the 16-bit table), the wrap as min(e, e - 2P) on unsigned values, an
            // absent term (j > deg or sigma_j = 0) parked on the zero entry behind the table (offset 2P, step 2P: min(4P, 2P) stays there),
            // sigma_0 a constant: five instructions per term and no branch, the table reads of a position in flight together. Before:
            // a compare, an EXEC mask, a branch and a full LDS wait per term, ~150 instructions per position (0.61 of the 0.92 ms that 4096
            // uncorrectable words of 8PSK 3/4 cost).
            auto chien = [&](auto nt_tag) {
                constexpr int NT = decltype(nt_tag)::value; // terms 1 .. NT (the code's t)
                const uint32_t P2 = 2u * P;
                uint32_t ex2[NT + 1], st2[NT + 1];
                const uint32_t i0 = (uint32_t)a.s + 1 + tid;
#pragma unroll
                for (int j = 1; j <= NT; j++) {
                    const uint32_t l = j <= deg ? lsig[j] : 0xffffffffu;
                    if (l == 0xffffffffu) { ex2[j] = P2; st2[j] = P2; }
                    else { st2[j] = 2u * modP((uint32_t)kBchThreads * (uint32_t)j, m, P); ex2[j] = 2u * modP(l + modP(i0 * (uint32_t)j, m, P), m, P); }
                }
                const uint32_t c0 = (deg >= 0 && lsig[0] != 0xffffffffu) ? (uint32_t)al[lsig[0]] : 0u;
                const uint8_t* alb = reinterpret_cast<const uint8_t*>(al);
                for (uint32_t i = i0; i <= (uint32_t)(n + a.s); i += kBchThreads) {
                    uint32_t res = c0;
#pragma unroll
                    for (int j = 1; j <= NT; j++) {
                        res ^= (uint32_t)*reinterpret_cast<const uint16_t*>(alb + ex2[j]);
                        const uint32_t e = ex2[j] + st2[j];
                        ex2[j] = min(e, e - P2);
                    }
                    if (res == 0) { const int idx = atomicAdd(&ctl[1], 1); if (idx < 16) roots[idx] = i; }
                }
            };
            if (t <= 8) chien(std::integral_constant<int, 8>{});
            else if (t <= 10) chien(std::integral_constant<int, 10>{});
            else chien(std::integral_constant<int, kMaxT>{});
            __syncthreads();
        }
        if (tid == 0) {
            int nr = ctl[1] < 16 ? ctl[1] : 16;
            if (mode == 2) {
                // ascending exponent order, at most deg of them; number = alpha^(P - exp), bit index = P - exp (mod P)
                for (int x = 1; x < nr; x++) { uint32_t v = roots[x]; int y = x - 1; while (y >= 0 && roots[y] > v) { roots[y + 1] = roots[y]; y--; } roots[y + 1] = v; }
                if (nr > deg) nr = deg;
                for (int x = 0; x < nr; x++) roots[x] = modP(P - modP(roots[x], m, P), m, P);
            }
            int status = (deg == nr) ? nr : -1;
            for (int x = 0; x < nr; x++) { // lib/bch.cc:429-452
                const uint32_t bit_idx = roots[x];
                if (bit_idx >= (uint32_t)n) { status = -2; break; } // "Error location number out of range" (throws)
                if (bit_idx < (uint32_t)(n - k)) continue;
                const uint32_t net = (uint32_t)n - 1 - bit_idx;
                out[net >> 3] ^= (uint8_t)(1u << (7 - (net & 7)));
            }
            a.corr[f] = status;
        }
    }
}

BchDecoderHip::BchDecoderHip(int m, uint32_t prim_poly, int t, int n, int max_frames, int device)
    : max_frames_(max_frames), device_(device)
{
    if (!code_.build(m, prim_poly, t, n, &err_)) return;
    if (code_.n % 8 || code_.k % 8) { err_ = "u8 array messages are only supported for n and k multiple of 8."; return; } // lib/bch.cc:19-24
    if (max_frames_ < 1 || max_frames_ > 65535) { err_ = "max_frames must be in 1..65535 (frames are one launch dimension)"; return; }
#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err_ = std::string(#x) + ": " + hipGetErrorString(e_); return; } } while (0)
    DeviceGuard dev_guard(device_); // restored on return (device_guard.h)
    if (!dev_guard.ok) { err_ = "hipSetDevice failed"; return; }
    hipDeviceProp_t pr;
    HIP_OK(hipGetDeviceProperties(&pr, device_));
    n_cus_ = pr.multiProcessorCount;
    HIP_OK(hipMalloc(&d_antilog_, code_.antilog.size() * 2));
    HIP_OK(hipMalloc(&d_log_, code_.log.size() * 2));
    HIP_OK(hipMalloc(&d_quad_, code_.quad.size() * 2));
    HIP_OK(hipMemcpy(d_antilog_, code_.antilog.data(), code_.antilog.size() * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_log_, code_.log.data(), code_.log.size() * 2, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_quad_, code_.quad.data(), code_.quad.size() * 2, hipMemcpyHostToDevice));
    {   // parity-check columns for the syndromes (bch_decode_kernel): per bit exponent e in [0, n) the t field elements alpha^((2u+1) e)
        std::vector<uint16_t> hc((size_t)code_.n * 16, 0);
        for (int e = 0; e < code_.n; e++)
            for (int u = 0; u < code_.t; u++) hc[(size_t)e * 16 + u] = code_.antilog[(uint32_t)(((uint64_t)(2 * u + 1) * (uint64_t)e) % (uint64_t)code_.P)];
        HIP_OK(hipMalloc(&d_hcol_, hc.size() * 2));
        HIP_OK(hipMemcpy(d_hcol_, hc.data(), hc.size() * 2, hipMemcpyHostToDevice));
    }
    {   // ... and by rows as 0 / 1 bytes for the batched product: row 16 u + b = bit b of alpha^((2u+1) e), column = stream position
        // p = n - 1 - e; rows padded to a multiple of 32, columns to a multiple of kSynK (the padding is zero: whatever the frames hold
        // behind their n bits does not count)
        synd_rt_ = (16 * code_.t + 31) / 32;
        synd_kp_ = (code_.n + kSynK - 1) / kSynK * kSynK;
        std::vector<int8_t> hr((size_t)synd_rt_ * 32 * synd_kp_, 0);
        for (int p = 0; p < code_.n; p++) {
            const uint64_t e = (uint64_t)(code_.n - 1 - p);
            for (int u = 0; u < code_.t; u++) {
                const uint32_t v = code_.antilog[(uint32_t)(((uint64_t)(2 * u + 1) * e) % (uint64_t)code_.P)];
                for (int b = 0; b < code_.m; b++) hr[(size_t)(16 * u + b) * synd_kp_ + p] = (int8_t)((v >> b) & 1u);
            }
        }
        HIP_OK(hipMalloc(&d_hrows_, hr.size()));
        HIP_OK(hipMemcpy(d_hrows_, hr.data(), hr.size(), hipMemcpyHostToDevice));
        HIP_OK(hipMalloc(&d_synd_, (size_t)max_frames_ * 32));
        synd_min_frames_ = 32;
        if (const char* ev = getenv("DVBS2_BCH_SYND_MIN")) synd_min_frames_ = std::max(1, atoi(ev)); // tests: 1 = always the product, 1000000 = never
    }
    lds_bytes_ = (((size_t)code_.P * 2 + 15) & ~(size_t)15) + kBchWorkWords * 4 + (size_t)((code_.n / 8 + 15) & ~15);
    HIP_OK(hipFuncSetAttribute((const void*)bch_decode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes_));
#undef HIP_OK
}

// lib/bbdescrambler_bb_impl.cc:51-65: PRBS 1 + x^14 + x^15, register loaded with 100101010000000, MSB-first bytes.
void bb_derandomise_sequence(uint8_t* seq, int n_bytes)
{
    uint32_t reg = 0x4A80;
    for (int i = 0; i < n_bytes; i++) {
        uint8_t v = 0;
        for (int bit = 7; bit >= 0; bit--) {
            const uint32_t fb = (reg ^ (reg >> 1)) & 1u;
            v |= (uint8_t)(fb << bit);
            reg = (reg >> 1) | (fb << 14);
        }
        seq[i] = v;
    }
}

int BchDecoderHip::set_descramble(bool enable)
{
    if (!ok()) return -1;
    call_err_.clear();
    DeviceGuard dev_guard(device_);
    if (!dev_guard.ok) { call_err_ = "hipSetDevice failed"; return -1; }
    if (enable && !d_scramble_) {
        std::vector<uint8_t> seq(code_.k / 8);
        bb_derandomise_sequence(seq.data(), (int)seq.size());
        if (hipMalloc(&d_scramble_, seq.size()) != hipSuccess ||
            hipMemcpy(d_scramble_, seq.data(), seq.size(), hipMemcpyHostToDevice) != hipSuccess) { call_err_ = "descramble sequence upload failed"; return -1; }
    }
    descramble_ = enable;
    return 0;
}

BchDecoderHip::~BchDecoderHip()
{
    DeviceGuard dev_guard(device_);
    (void)hipFree(d_scramble_);
    (void)hipFree(d_antilog_); (void)hipFree(d_log_); (void)hipFree(d_quad_); (void)hipFree(d_hcol_); (void)hipFree(d_hrows_); (void)hipFree(d_synd_);
    for (InFlight& t : track_) if (t.done) (void)hipEventDestroy(t.done);
}

int BchDecoderHip::decode_device(const uint8_t* d_cw, int n_frames, uint8_t* d_msg, int32_t* d_corr, hipStream_t stream,
                                 const uint8_t* d_llr_state, int llr_stride, int frame_base)
{
    if (!ok()) return -1;
    call_err_.clear();
    if (n_frames < 0 || frame_base < 0 || frame_base + n_frames > max_frames_) { call_err_ = "n_frames exceeds max_frames"; return -1; }
    if (n_frames == 0) return 0;
    DeviceGuard dev_guard(device_);
    if (!dev_guard.ok) { call_err_ = "hipSetDevice failed"; return -1; }
    // the syndrome words of [frame_base, frame_base + n_frames) belong to this call until its per-frame kernel has read them: an earlier
    // call on ANOTHER stream that used an overlapping range is waited for on the device (same stream: stream order already does it)
    for (const InFlight& t : track_)
        if (t.done && t.stream != stream && t.base < frame_base + n_frames && frame_base < t.base + t.n)
            if (hipStreamWaitEvent(stream, t.done, 0) != hipSuccess) { call_err_ = "hipStreamWaitEvent failed"; return -1; }
    uint32_t* const synd = d_synd_ + (size_t)frame_base * 8;
    BchArgs a;
    a.antilog = d_antilog_; a.log = d_log_; a.quad = d_quad_; a.hcol = reinterpret_cast<const uint4*>(d_hcol_); a.cw = d_cw; a.msg = d_msg; a.corr = d_corr; a.llr_state = d_llr_state; a.llr_stride = llr_stride;
    a.descramble = descramble_ ? d_scramble_ : nullptr;
    a.n_frames = n_frames; a.m = code_.m; a.P = code_.P; a.t = code_.t; a.n = code_.n; a.k = code_.k; a.s = code_.s;
    a.synd = nullptr;
    if (n_frames >= synd_min_frames_ && (d_cw != nullptr || (llr_stride >= synd_kp_ && llr_stride % 8 == 0))) {
        // the odd syndromes of the whole batch first (bch_syndrome_kernel): 32 frames per wave, the columns cut into chunks so that the
        // launch has about eight waves per CU
        const int tiles = (n_frames + 31) / 32, steps = synd_kp_ / kSynK;
        const int chunks = std::max(1, std::min(steps, (8 * std::max(1, n_cus_) + tiles - 1) / tiles));
        const int spc = (steps + chunks - 1) / chunks;
        const dim3 sgrid((tiles + 3) / 4, (steps + spc - 1) / spc);
        if (hipMemsetAsync(synd, 0, (size_t)n_frames * 32, stream) != hipSuccess) { call_err_ = "bch syndrome buffer reset failed"; return -1; }
        const bool packed = d_cw != nullptr;
        const uint8_t* src = packed ? d_cw : d_llr_state;
        const size_t stride = packed ? (size_t)(code_.n / 8) : (size_t)llr_stride;
#define DVBS2_SYN_LAUNCH(RT) do { \
            if (packed) hipLaunchKernelGGL((bch_syndrome_kernel<RT, true>), sgrid, dim3(256), 0, stream, d_hrows_, synd_kp_, src, stride, code_.n / 8, n_frames, spc, synd); \
            else hipLaunchKernelGGL((bch_syndrome_kernel<RT, false>), sgrid, dim3(256), 0, stream, d_hrows_, synd_kp_, src, stride, code_.n / 8, n_frames, spc, synd); } while (0)
        switch (synd_rt_) {
            case 1: DVBS2_SYN_LAUNCH(1); break; case 2: DVBS2_SYN_LAUNCH(2); break; case 3: DVBS2_SYN_LAUNCH(3); break;
            case 4: DVBS2_SYN_LAUNCH(4); break; case 5: DVBS2_SYN_LAUNCH(5); break; default: DVBS2_SYN_LAUNCH(6); break;
        }
#undef DVBS2_SYN_LAUNCH
        a.synd = synd;
    }
    const int grid = std::min(n_frames, std::max(1, n_cus_));
    hipLaunchKernelGGL(bch_decode_kernel, dim3(grid), dim3(kBchThreads), lds_bytes_, stream, a);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { call_err_ = std::string("bch kernel launch: ") + hipGetErrorString(e); return -1; }
    {
        InFlight& t = track_[track_next_];
        track_next_ = (track_next_ + 1) % kTrack;
        if (!t.done && hipEventCreateWithFlags(&t.done, hipEventDisableTiming) != hipSuccess) { t.done = nullptr; call_err_ = "hipEventCreate failed"; (void)hipStreamSynchronize(stream); return -1; }
        t.stream = stream; t.base = frame_base; t.n = n_frames;
        if (hipEventRecord(t.done, stream) != hipSuccess) { call_err_ = "hipEventRecord failed"; (void)hipStreamSynchronize(stream); return -1; }
    }
    return 0;
}

} // namespace dvbs2
