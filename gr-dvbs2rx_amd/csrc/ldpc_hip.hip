// ldpc_hip.hip -- layered offset-min-sum int8 LDPC belief propagation for DVB-S2/S2X/T2 on gfx950.
//
// What it computes (bit-exact contract): LDPCDecoder<SIMD<int8_t,W>, OffsetMinSumAlgorithm<...,
// NormalUpdate, FACTOR 2>>::operator() of the reference (lib/ldpc_decoder/layered_decoder.hh:143-160,
// lib/ldpc_decoder/algorithms.hh:151-207) for every frame, with the batch-coupled stopping rule of a
// G-frame SIMD batch, followed by the hard decision + MSB-first packing of
// ldpc_decoder_bb_impl::general_work (lib/ldpc_decoder_bb_impl.cc:432-442).
//
// Mapping (MI355X-first, not a translation of the lane-per-frame SIMD decoder):
//   * one FECFRAME per workgroup of 6 wavefronts; thread j (< 360) owns check node (layer i, lane j)
//     of every layer, i.e. one row of each circulant;
//   * the frame's N LLRs live in LDS for the whole decode (offset-binary bytes, parity part permuted
//     to [layer][lane]); a table entry (group g, shift s) is a rotated contiguous 360-byte window of it;
//   * the check-to-bit messages are private to the owning thread: they stream through L2/Infinity
//     Cache as coalesced dwords (4 int8 messages per dword), never through LDS;
//   * layers with two entries of the same group (sequential-order hazard of the reference's strictly
//     ordered update) are processed in ascending lane blocks of B_i (ldpc_schedule.h).
#include "ldpc_hip.h"
#include <algorithm>
#include <cstdio>
#include <vector>

namespace dvbs2 {

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err_ = std::string(#x) + ": " + hipGetErrorString(e_); return; } } while (0)
#define HIP_RET(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { err_ = std::string(#x) + ": " + hipGetErrorString(e_); return -1; } } while (0)

constexpr int kThreads = 384; // 6 wavefronts; lanes 0..359 own the 360 rows of a circulant
constexpr int kM = 360;

struct LdpcKernelArgs {
    const uint32_t* layers;  // per layer: entry_off, cnt | block << 16
    const uint32_t* entries; // base | rot << 16
    const int8_t* llr_in;    // fresh frames (natural order) or nullptr when resuming
    uint8_t* state;          // internal layout, offset binary
    uint32_t* msgs;
    int* iters;              // per frame, in/out
    int* good;               // per frame, out
    const int* target;       // per frame (resume) or nullptr (use cap)
    int N, K, q, wpc, cap, stop_on_good;
};

template <int DMAX>
__global__ __launch_bounds__(kThreads) void ldpc_layered_kernel(LdpcKernelArgs a)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
    const int f = blockIdx.x;
    const int tid = threadIdx.x;
    const int N = a.N, K = a.K, q = a.q;

    int it;
    const int tgt = a.target ? a.target[f] : a.cap;
    if (a.llr_in) {
        it = 0;
        const int8_t* src = a.llr_in + (size_t)f * N;
        for (int n = tid; n < N; n += kThreads) {
            uint8_t v = (uint8_t)src[n] ^ 0x80u;
            if (n < K) lds[n] = v;
            else { int r = n - K; lds[K + kM * (r % q) + r / q] = v; } // pty[360*i+j] = parity[q*j+i]
        }
    } else {
        it = a.iters[f];
        if (it >= tgt) return; // nothing to do for this frame in this pass (uniform)
        const uint8_t* src = a.state + (size_t)f * N;
        for (int n = tid; n < N; n += kThreads) lds[n] = src[n];
    }
    __syncthreads();

    uint32_t* msg_base = a.msgs + (size_t)f * q * a.wpc * kThreads;
    const int j = tid;
    bool good = false;

    for (;;) {
        if (a.stop_on_good || it >= tgt) {
            // ---- syndrome test: layered_decoder.hh:32-49, algorithms.hh:195-202 (cnv <= 0 is bad:
            // a zero LLR on a check or an odd number of negative LLRs) ----
            int bad = 0;
            if (j < kM) {
                for (int i = 0; i < q; i++) {
                    const uint32_t eoff = a.layers[2 * i];
                    const int deg = (int)(a.layers[2 * i + 1] & 0xffff) + 2;
                    uint32_t x = 0, z = 0;
                    for (int k = 0; k < deg; k++) {
                        const uint32_t e = a.entries[eoff + k];
                        int t = j + (int)(e >> 16);
                        t = t >= kM ? t - kM : t;
                        uint32_t v = lds[(e & 0xffff) + t];
                        const bool valid = (i | j | (k ^ (deg - 1))) != 0; // check (0,0) has no previous parity
                        v = valid ? v : 0x81u; // +1: neutral
                        x ^= v;
                        z |= (v == 0x80u);
                    }
                    // offset binary: sign bit is inverted; deg links (deg-1 real + neutral for (0,0))
                    const uint32_t neg_parity = ((x >> 7) ^ (uint32_t)deg) & 1u;
                    bad |= (int)(neg_parity | z);
                }
            }
            good = !__syncthreads_or(bad);
        }
        if (it >= tgt) break;
        if (a.stop_on_good && good) break;

        // ---- one update sweep: layered_decoder.hh:50-79 ----
        const bool first = (it == 0); // bnl == 0 (layered_decoder.hh:27-31)
        for (int i = 0; i < q; i++) {
            const uint32_t eoff = a.layers[2 * i];
            const uint32_t cb = a.layers[2 * i + 1];
            const int deg = (int)(cb & 0xffff) + 2;
            const int block = (int)(cb >> 16);
            const int nw = (deg + 3) >> 2;
            uint32_t* mp = msg_base + (size_t)i * a.wpc * kThreads;
            for (int start = 0; start < kM; start += block) {
                const int jj = start + tid; // block >= 360 -> single pass with jj = tid
                if (tid < block && jj < kM) {
                    uint32_t mw[DMAX / 4];
#pragma unroll
                    for (int w = 0; w < DMAX / 4; w++) mw[w] = (w < nw && !first) ? mp[w * kThreads + jj] : 0x80808080u;
                    int d[DMAX], mg[DMAX], ad[DMAX];
                    int min0 = 127, min1 = 127, signs = 0;
#pragma unroll
                    for (int k = 0; k < DMAX; k++) {
                        if (k < deg) {
                            const uint32_t e = a.entries[eoff + k];
                            int t = jj + (int)(e >> 16);
                            t = t >= kM ? t - kM : t;
                            ad[k] = (int)(e & 0xffff) + t;
                            const int Lb = lds[ad[k]];
                            const int mb = (int)((mw[k >> 2] >> (8 * (k & 3))) & 0xffu);
                            const bool valid = (i | jj | (k ^ (deg - 1))) != 0;
                            // R1: inp = sat8(L - m); R2: mag = usat(qabs(inp) - 1) = med3(|L - m| - 1, 0, 126)
                            int dd = Lb - mb;
                            int ab = dd < 0 ? -dd : dd;
                            int mag = min(max(ab - 1, 0), 126);
                            dd = valid ? dd : 0;
                            mag = valid ? mag : 127;
                            d[k] = dd; mg[k] = mag;
                            // R3: two smallest magnitudes; R4: xor of signs
                            min1 = min(min1, max(min0, mag));
                            min0 = min(min0, mag);
                            signs ^= dd;
                        }
                    }
                    uint32_t nm[DMAX / 4];
#pragma unroll
                    for (int w = 0; w < DMAX / 4; w++) nm[w] = 0;
#pragma unroll
                    for (int k = 0; k < DMAX; k++) {
                        if (k < deg) {
                            const bool valid = (i | jj | (k ^ (deg - 1))) != 0;
                            // R5: out = vsign(other, (signs ^ x) | 127)
                            const int other = (mg[k] == min0) ? min1 : min0;
                            const int out = ((signs ^ d[k]) < 0) ? -other : other;
                            // R6: LLR = sat8(inp + out), unclamped out
                            const int inp = min(max(d[k], -128), 127);
                            const int nl = min(max(inp + out, -128), 127) + 128;
                            if (valid) lds[ad[k]] = (uint8_t)nl;
                            // R7: bnl = clamp(out, -32, 31)
                            const int nmsg = min(max(out, -32), 31) + 128;
                            nm[k >> 2] |= (uint32_t)nmsg << (8 * (k & 3));
                        }
                    }
#pragma unroll
                    for (int w = 0; w < DMAX / 4; w++)
                        if (w < nw) mp[w * kThreads + jj] = nm[w];
                }
                __syncthreads();
            }
        }
        it++;
    }

    if (tid == 0) { a.iters[f] = it; a.good[f] = good ? 1 : 0; }
    uint8_t* dst = a.state + (size_t)f * N;
    for (int n = tid; n < N; n += kThreads) dst[n] = lds[n];
}

// Per-group stopping rule of one reference SIMD batch: while (bad(any lane) && --trials >= 0) update(all lanes)
// (layered_decoder.hh:153). After a pass, every frame f sits at iters[f] updates with good[f] known there.
__global__ void ldpc_group_targets_kernel(const int* iters, const int* good, int* target, int* flag,
                                          int* ret, int n_groups, int G, int n_frames, int cap)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= n_groups) return;
    const int f0 = g * G, f1 = min(f0 + G, n_frames);
    int T = 0;
    for (int f = f0; f < f1; f++) T = max(T, iters[f]);
    bool aligned = true, all_good = true;
    for (int f = f0; f < f1; f++) {
        if (iters[f] != T) aligned = false;
        else if (!good[f]) all_good = false;
    }
    int tgt = T;
    bool unresolved = false;
    if (!aligned) unresolved = true;               // bring the early stoppers up to T, then look again
    else if (!all_good && T < cap) { tgt = T + 1; unresolved = true; }
    for (int f = f0; f < f1; f++) target[f] = tgt;
    if (unresolved) atomicAdd(flag, 1);
    else if (ret) ret[g] = all_good ? cap - T : -1;
}

// Hard decision + MSB-first packing (ldpc_decoder_bb_impl.cc:432-442) and optional soft output
// (the llr_pdu payload, :422-429), undoing the parity permutation (layered_decoder.hh:155-157).
__global__ void ldpc_finalize_kernel(const uint8_t* state, uint8_t* bits, int8_t* llr_out,
                                     int N, int K, int q, int out_bytes)
{
    const int f = blockIdx.y;
    const uint8_t* s = state + (size_t)f * N;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= N / 8) return;
    uint32_t byte = 0;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int n = 8 * b + k;
        int idx = n;
        if (n >= K) { const int r = n - K; idx = K + kM * (r % q) + r / q; }
        const uint32_t v = s[idx];
        if (v < 0x80u) byte |= 1u << (7 - k);
        if (llr_out) llr_out[(size_t)f * N + n] = (int8_t)(v ^ 0x80u);
    }
    if (b < out_bytes) bits[(size_t)f * out_bytes + b] = (uint8_t)byte;
}

LdpcDecoderHip::LdpcDecoderHip(const LdpcTableDesc* table, int out_bits_message, int group_size, int max_frames, int device)
    : out_bits_message_(out_bits_message), G_(group_size), max_frames_(max_frames), device_(device)
{
    if (!compile_ldpc_schedule(table, &sched_)) { err_ = "unknown or inconsistent LDPC table"; return; }
    if (G_ < 1 || max_frames_ < 1) { err_ = "bad group_size/max_frames"; return; }
    if (out_bits_message_ <= 0 || out_bits_message_ > sched_.N || out_bits_message_ % 8) { err_ = "bad message length"; return; }
    const int degmax = sched_.cnt_max + 2;
    dmax_ = degmax <= 8 ? 8 : degmax <= 16 ? 16 : 32;
    if (degmax > 32) { err_ = "check degree > 32 unsupported"; return; }
    words_per_check_ = (degmax + 3) / 4;
    HIP_OK(hipSetDevice(device_));
    std::vector<uint32_t> hl(2 * sched_.q), he(sched_.entries.size());
    for (int i = 0; i < sched_.q; i++) {
        hl[2 * i] = sched_.layers[i].entry_off;
        hl[2 * i + 1] = sched_.layers[i].cnt | ((uint32_t)sched_.layers[i].block << 16);
    }
    for (size_t i = 0; i < he.size(); i++) he[i] = sched_.entries[i].base | ((uint32_t)sched_.entries[i].rot << 16);
    HIP_OK(hipMalloc(&d_layers_, hl.size() * 4));
    HIP_OK(hipMalloc(&d_entries_, he.size() * 4));
    HIP_OK(hipMemcpy(d_layers_, hl.data(), hl.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_entries_, he.data(), he.size() * 4, hipMemcpyHostToDevice));
    HIP_OK(hipMalloc(&d_state_, (size_t)max_frames_ * sched_.N));
    HIP_OK(hipMalloc(&d_msgs_, (size_t)max_frames_ * sched_.q * words_per_check_ * kThreads * 4));
    HIP_OK(hipMalloc(&d_iters_, (size_t)max_frames_ * 4));
    HIP_OK(hipMalloc(&d_good_, (size_t)max_frames_ * 4));
    HIP_OK(hipMalloc(&d_target_, (size_t)max_frames_ * 4));
    HIP_OK(hipMalloc(&d_flag_, 4));
    HIP_OK(hipHostMalloc(&h_flag_, 4));
    HIP_OK(hipEventCreate(&ev0_));
    HIP_OK(hipEventCreate(&ev1_));
    const size_t lds_bytes = (size_t)sched_.N;
    const void* fn = dmax_ == 8 ? (const void*)ldpc_layered_kernel<8> : dmax_ == 16 ? (const void*)ldpc_layered_kernel<16> : (const void*)ldpc_layered_kernel<32>;
    HIP_OK(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
}

LdpcDecoderHip::~LdpcDecoderHip()
{
    (void)hipSetDevice(device_);
    (void)hipFree(d_layers_); (void)hipFree(d_entries_); (void)hipFree(d_state_); (void)hipFree(d_msgs_);
    (void)hipFree(d_iters_); (void)hipFree(d_good_); (void)hipFree(d_target_); (void)hipFree(d_flag_);
    if (h_flag_) (void)hipHostFree(h_flag_);
    if (ev0_) (void)hipEventDestroy(ev0_);
    if (ev1_) (void)hipEventDestroy(ev1_);
}

int LdpcDecoderHip::decode_device(const int8_t* d_llr_in, int n_frames, int max_trials, int out_mode,
                                  uint8_t* d_bits_out, int8_t* d_llr_out, int32_t* d_ret, hipStream_t stream)
{
    if (!ok()) return -1;
    if (n_frames < 0 || n_frames > max_frames_) { err_ = "n_frames exceeds max_frames"; return -1; }
    if (n_frames == 0) return 0;
    if (max_trials < 0) { err_ = "max_trials < 0"; return -1; }
    HIP_RET(hipSetDevice(device_));
    LdpcKernelArgs a;
    a.layers = d_layers_; a.entries = d_entries_; a.llr_in = d_llr_in; a.state = d_state_; a.msgs = d_msgs_;
    a.iters = d_iters_; a.good = d_good_; a.target = nullptr;
    a.N = sched_.N; a.K = sched_.K; a.q = sched_.q; a.wpc = words_per_check_; a.cap = max_trials; a.stop_on_good = 1;
    const size_t lds_bytes = (size_t)sched_.N;
    auto launch = [&](const LdpcKernelArgs& ka) {
        if (profiling_) (void)hipEventRecord(ev0_, stream);
        if (dmax_ == 8) hipLaunchKernelGGL(ldpc_layered_kernel<8>, dim3(n_frames), dim3(kThreads), lds_bytes, stream, ka);
        else if (dmax_ == 16) hipLaunchKernelGGL(ldpc_layered_kernel<16>, dim3(n_frames), dim3(kThreads), lds_bytes, stream, ka);
        else hipLaunchKernelGGL(ldpc_layered_kernel<32>, dim3(n_frames), dim3(kThreads), lds_bytes, stream, ka);
        if (profiling_) {
            (void)hipEventRecord(ev1_, stream);
            (void)hipEventSynchronize(ev1_);
            float ms = 0; (void)hipEventElapsedTime(&ms, ev0_, ev1_);
            prof_ms_ += ms; prof_launches_++;
        }
    };
    launch(a);
    HIP_RET(hipGetLastError());
    const int n_groups = (n_frames + G_ - 1) / G_;
    for (int round = 0;; round++) {
        HIP_RET(hipMemsetAsync(d_flag_, 0, 4, stream));
        hipLaunchKernelGGL(ldpc_group_targets_kernel, dim3((n_groups + 127) / 128), dim3(128), 0, stream,
                           d_iters_, d_good_, d_target_, d_flag_, d_ret, n_groups, G_, n_frames, max_trials);
        HIP_RET(hipMemcpyAsync(h_flag_, d_flag_, 4, hipMemcpyDeviceToHost, stream));
        HIP_RET(hipStreamSynchronize(stream));
        if (*h_flag_ == 0) break;
        if (round > 2 * max_trials + 2) { err_ = "group resolution did not converge"; return -1; }
        LdpcKernelArgs r = a;
        r.llr_in = nullptr; r.target = d_target_; r.stop_on_good = 0;
        launch(r);
        HIP_RET(hipGetLastError());
    }
    const int out_bytes = (out_mode ? out_bits_message_ : sched_.N) / 8;
    hipLaunchKernelGGL(ldpc_finalize_kernel, dim3((sched_.N / 8 + 255) / 256, n_frames), dim3(256), 0, stream,
                       d_state_, d_bits_out, d_llr_out, sched_.N, sched_.K, sched_.q, out_bytes);
    HIP_RET(hipGetLastError());
    HIP_RET(hipStreamSynchronize(stream));
    return 0;
}

} // namespace dvbs2
