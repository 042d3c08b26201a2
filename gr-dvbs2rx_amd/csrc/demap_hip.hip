// demap_hip.hip -- soft demapper kernels. Streaming, HBM-bound: 8 bytes in per symbol, n_mod bytes out.
//
// Arithmetic restated (all float, no FMA contraction: the file is compiled with -ffp-contract=off and the
// products below use explicit round-to-nearest intrinsics):
//   QPSK  lib/qpsk.h:208-214: scalar = (float)(2*sqrt(2) / N0); out = sat8(rint(x * scalar)) over the 2*n_syms
//         floats (volk_32f_s32f_convert_8i; VOLK is not part of the reference tree -- see oracle/demap_oracle.c).
//   8PSK  lib/psk.hh:143-150 with quantize :123-131 and rot :113; precision = (float)(4.0 / N0)
//         (lib/xfecframe_demapper_cb_impl.cc:148); column de-interleave :162-176.
#include "demap_hip.h"
#include "demap_math.hpp"
#include <cmath>
#include "../../include/dvbs2_fec_hip.h"

#include "device_guard.h"
namespace dvbs2 {

__global__ void demap_qpsk_kernel(const float4* __restrict__ syms, const float* __restrict__ n0, int n0_count,
                                  uint32_t* __restrict__ out, int quads_per_frame, int n_frames)
{
    const int f = blockIdx.y;
    const float N0 = n0[n0_count > 1 ? f : 0];
    const float scalar = qpsk_scalar(N0);
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < quads_per_frame; i += gridDim.x * blockDim.x) {
        const float4 v = syms[(size_t)f * quads_per_frame + i]; // two symbols
        const uint32_t b0 = (uint8_t)qpsk_llr(v.x, scalar), b1 = (uint8_t)qpsk_llr(v.y, scalar);
        const uint32_t b2 = (uint8_t)qpsk_llr(v.z, scalar), b3 = (uint8_t)qpsk_llr(v.w, scalar);
        out[(size_t)f * quads_per_frame + i] = b0 | (b1 << 8) | (b2 << 16) | (b3 << 24);
    }
}

// One thread per FOUR consecutive symbols: two 16-byte loads, and one dword store into each of the three de-interleaved
// columns (rows = n_syms is a multiple of 4 for every frame size, so the column bases are dword aligned). One thread per
// symbol with three byte stores reached 3.9 TB/s of algorithmic bytes; a wave now stores 256 contiguous bytes per instruction.
__global__ void demap_8psk_kernel(const float4* __restrict__ syms, const float* __restrict__ n0, int n0_count,
                                  uint32_t* __restrict__ out, int n_quads, int ra0, int ra1, int ra2, float rr, float ri)
{
    const int f = blockIdx.y;
    const float N0 = n0[n0_count > 1 ? f : 0];
    const float dp = psk8_dist_prec(N0);
    uint32_t* o = out + (size_t)f * 3 * n_quads; // 3 * n_syms bytes per frame
    for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n_quads; j += gridDim.x * blockDim.x) {
        const float4 a = syms[((size_t)f * n_quads + j) * 2], b = syms[((size_t)f * n_quads + j) * 2 + 1];
        const float re[4] = { a.x, a.z, b.x, b.z }, im[4] = { a.y, a.w, b.y, b.w };
        uint32_t w0 = 0, w1 = 0, w2 = 0;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            int8_t b0, b1, b2;
            psk8_llr(re[k], im[k], rr, ri, dp, b0, b1, b2);
            w0 |= (uint32_t)(uint8_t)b0 << (8 * k);
            w1 |= (uint32_t)(uint8_t)b1 << (8 * k);
            w2 |= (uint32_t)(uint8_t)b2 << (8 * k);
        }
        o[ra1 / 4 + j] = w1;
        o[ra2 / 4 + j] = w2;
        o[ra0 / 4 + j] = w0;
    }
}

// One workgroup per frame. llr == nullptr: reference point = hard slice of the symbol (pre-decoder estimate);
// otherwise the reference point is re-mapped from the signs of the decoded LLRs (post-decoder refinement,
// lib/xfecframe_demapper_cb_impl.cc:268-307, lib/qpsk.h:266-281): LLR < 0 -> -1, else +1; 8PSK bits are picked
// through the column interleaver (ra0..2 = d_rowaddr0..2).
__global__ void demap_snr_kernel(const float2* __restrict__ syms, const int8_t* __restrict__ llr, float* __restrict__ snr,
                                 int n_syms, int constellation, int ra0, int ra1, int ra2, float rr, float ri)
{
    __shared__ float ssp[256], snp[256];
    const int f = blockIdx.x, tid = threadIdx.x;
    const float rs2 = 0.70710678118654752440f;
    float sp = 0, np = 0;
    for (int j = tid; j < n_syms; j += blockDim.x) {
        const float2 c = syms[(size_t)f * n_syms + j];
        float sr, si;
        if (constellation == DVBS2_MOD_QPSK) {
            if (llr) {
                const int8_t* l = llr + (size_t)f * 2 * n_syms + 2 * j;
                sr = l[0] >= 0 ? rs2 : -rs2; si = l[1] >= 0 ? rs2 : -rs2;
            } else { sr = c.x >= 0 ? rs2 : -rs2; si = c.y >= 0 ? rs2 : -rs2; }
        } else {
            int b0, b1, b2;
            if (llr) {
                const int8_t* l = llr + (size_t)f * 3 * n_syms;
                b0 = l[ra0 + j] < 0 ? -1 : 1; b1 = l[ra1 + j] < 0 ? -1 : 1; b2 = l[ra2 + j] < 0 ? -1 : 1;
            } else {
                const float cr = c.x * rr - c.y * ri, ci = c.x * ri + c.y * rr;
                b1 = cr < 0 ? -1 : 1; b2 = ci < 0 ? -1 : 1; b0 = fabsf(cr) < fabsf(ci) ? -1 : 1;
            }
            const int idx = (((b0 + 1) << 1) ^ 0x4) | ((b1 + 1) ^ 0x2) | (((b2 + 1) >> 1) ^ 0x1);
            const float m8r[8] = { rs2, 1, -1, -rs2, 0, rs2, -rs2, 0 }, m8i[8] = { rs2, 0, 0, -rs2, 1, -rs2, rs2, -1 };
            sr = m8r[idx]; si = m8i[idx];
        }
        const float er = c.x - sr, ei = c.y - si;
        sp += sr * sr + si * si; np += er * er + ei * ei;
    }
    ssp[tid] = sp; snp[tid] = np;
    __syncthreads();
    for (int s = 128; s; s >>= 1) { if (tid < s) { ssp[tid] += ssp[tid + s]; snp[tid] += snp[tid + s]; } __syncthreads(); }
    if (tid == 0) { float n = snp[0]; if (!(n > 0)) n = 1e-12f; snr[f] = ssp[0] / n; }
}

DemapperHip::DemapperHip(int framesize, int rate, int constellation, int max_frames, int device)
    : constellation_(constellation), max_frames_(max_frames), device_(device)
{
    n_llr_ = framesize == DVBS2_FECFRAME_NORMAL ? 64800 : framesize == DVBS2_FECFRAME_MEDIUM ? 32400 : 16200;
    if (constellation == DVBS2_MOD_QPSK) n_mod_ = 2;
    else if (constellation == DVBS2_MOD_8PSK) {
        n_mod_ = 3;
        // rate enumerators: C3_5 = 4; C25_36 = 26, C13_18 = 28, C7_15 = 38, C8_15 = 39, C26_45 = 19 (dvb_config.h:20-72)
        if (rate == 4) order_ = 1;                                                              // "210"
        else if (rate == 26 || rate == 28 || rate == 38 || rate == 39 || rate == 19) order_ = 2; // "102"
        else order_ = 0;                                                                         // "012"
    } else { err_ = "Unsupported constellation"; return; }
    if (max_frames_ < 1 || max_frames_ > 65535) { err_ = "max_frames must be in 1..65535 (frames are one launch dimension)"; return; }
}

int DemapperHip::soft_device(const float* d_syms, int n_frames, const float* d_n0, int n0_count, int8_t* d_llr, hipStream_t stream)
{
    if (!ok()) return -1;
    call_err_.clear();
    if (n_frames < 0 || n_frames > max_frames_) { call_err_ = "n_frames exceeds max_frames"; return -1; }
    if (n_frames == 0) return 0;
    DeviceGuard dev_guard(device_);
    if (!dev_guard.ok) { call_err_ = "hipSetDevice failed"; return -1; }
    if (constellation_ == DVBS2_MOD_QPSK) {
        const int quads = n_llr_ / 4;
        hipLaunchKernelGGL(demap_qpsk_kernel, dim3((quads + 255) / 256, n_frames), dim3(256), 0, stream,
                           reinterpret_cast<const float4*>(d_syms), d_n0, n0_count, reinterpret_cast<uint32_t*>(d_llr), quads, n_frames);
    } else {
        const int rows = n_syms();
        int ra0 = 0, ra1 = rows, ra2 = 2 * rows;
        if (order_ == 1) { ra0 = 2 * rows; ra1 = rows; ra2 = 0; }
        else if (order_ == 2) { ra0 = rows; ra1 = 0; ra2 = 2 * rows; }
        const float rr = (float)std::cos(-M_PI / 8), ri = (float)std::sin(-M_PI / 8); // (complexf) exp(-j pi/8)
        const int quads = rows / 4; // 21600, 10800, 5400 symbols: a multiple of 4
        hipLaunchKernelGGL(demap_8psk_kernel, dim3((quads + 255) / 256, n_frames), dim3(256), 0, stream,
                           reinterpret_cast<const float4*>(d_syms), d_n0, n0_count, reinterpret_cast<uint32_t*>(d_llr), quads, ra0, ra1, ra2, rr, ri);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { call_err_ = std::string("demap kernel launch: ") + hipGetErrorString(e); return -1; }
    return 0;
}

DemapFused DemapperHip::fused(const float* d_syms, const float* d_n0, int n0_count) const
{
    DemapFused d{};
    d.syms = d_syms; d.n0 = d_n0; d.n0_count = n0_count; d.n_syms = n_syms();
    d.mode = constellation_ == DVBS2_MOD_QPSK ? 1 : 2;
    const int rows = n_syms();
    d.ra0 = 0; d.ra1 = rows; d.ra2 = 2 * rows;
    if (order_ == 1) { d.ra0 = 2 * rows; d.ra1 = rows; d.ra2 = 0; }
    else if (order_ == 2) { d.ra0 = rows; d.ra1 = 0; d.ra2 = 2 * rows; }
    d.rr = (float)std::cos(-M_PI / 8); d.ri = (float)std::sin(-M_PI / 8);
    return d;
}

int DemapperHip::snr_device(const float* d_syms, const int8_t* d_ref_llr, int n_frames, float* d_snr, hipStream_t stream)
{
    if (!ok()) return -1;
    call_err_.clear();
    if (n_frames < 0 || n_frames > max_frames_) { call_err_ = "n_frames exceeds max_frames"; return -1; }
    if (n_frames == 0) return 0;
    DeviceGuard dev_guard(device_);
    if (!dev_guard.ok) { call_err_ = "hipSetDevice failed"; return -1; }
    const float rr = (float)std::cos(-M_PI / 8), ri = (float)std::sin(-M_PI / 8);
    const int rows = n_syms();
    int ra0 = 0, ra1 = rows, ra2 = 2 * rows;
    if (order_ == 1) { ra0 = 2 * rows; ra1 = rows; ra2 = 0; }
    else if (order_ == 2) { ra0 = rows; ra1 = 0; ra2 = 2 * rows; }
    hipLaunchKernelGGL(demap_snr_kernel, dim3(n_frames), dim3(256), 0, stream,
                       reinterpret_cast<const float2*>(d_syms), d_ref_llr, d_snr, rows, constellation_, ra0, ra1, ra2, rr, ri);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { call_err_ = std::string("snr kernel launch: ") + hipGetErrorString(e); return -1; }
    return 0;
}

} // namespace dvbs2
