// demap_math.hpp -- the soft-demapper arithmetic as device functions, shared by the stand-alone demapper kernels (demap_hip.hip)
// and by the LDPC sweep kernels, which can take XFECFRAME symbols directly and demap while they load a frame into LDS
// (one launch and 2 N bytes of HBM traffic per frame less than demapper -> LLR buffer -> decoder).
// Arithmetic restated (all float, no FMA contraction: compiled with -ffp-contract=off, products by explicit round-to-nearest
// intrinsics):
//   QPSK  lib/qpsk.h:208-214: scalar = (float)(2*sqrt(2) / N0); out = sat8(rint(x * scalar)) (volk_32f_s32f_convert_8i; VOLK is
//         not part of the reference tree -- see oracle/demap_oracle.c).
//   8PSK  lib/psk.hh:143-150 with quantize :123-131 and rot :113; precision = (float)(4.0 / N0)
//         (lib/xfecframe_demapper_cb_impl.cc:148); column de-interleave :162-176.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

namespace dvbs2 {

__device__ __forceinline__ int8_t sat8_rint(float v)
{
    if (v > 127.0f) return 127;
    if (v < -128.0f) return -128;
    return (int8_t)rintf(v);
}
__device__ __forceinline__ float qpsk_scalar(float N0) { return (float)(2.0 * 1.41421356237309504880 / (double)N0); }
__device__ __forceinline__ int8_t qpsk_llr(float x, float scalar) { return sat8_rint(__fmul_rn(x, scalar)); }

__device__ __forceinline__ int8_t quant8(float dist_prec, float value)
{
    value = __fmul_rn(value, dist_prec);
    value = rintf(value);
    value = fminf(fmaxf(value, -128.0f), 127.0f);
    return (int8_t)value;
}
__device__ __forceinline__ float psk8_dist_prec(float N0)
{
    const float precision = (float)(4.0 / (double)N0);
    const float sin_pi_8 = 0.38268343236508977173f;
    return __fmul_rn(2 * sin_pi_8, precision);
}
// the three LLRs of one 8PSK symbol in the order soft[0], soft[1], soft[2] of PhaseShiftKeying<8>::soft
__device__ __forceinline__ void psk8_llr(float re, float im, float rr, float ri, float dp, int8_t& b0, int8_t& b1, int8_t& b2)
{
    const float rcp_sqrt_2 = 0.70710678118654752440f;
    const float cr = __fsub_rn(__fmul_rn(re, rr), __fmul_rn(im, ri));
    const float ci = __fadd_rn(__fmul_rn(re, ri), __fmul_rn(im, rr));
    b1 = quant8(dp, cr);
    b2 = quant8(dp, ci);
    b0 = quant8(dp, __fmul_rn(rcp_sqrt_2, __fsub_rn(fabsf(cr), fabsf(ci))));
}

// What a sweep kernel needs to demap while loading (passed by value; mode 0 = LLR input)
struct DemapFused {
    const float* syms;  // n_frames * n_syms complex symbols (re, im)
    const float* n0;    // one value, or one per frame
    int n0_count;
    int mode;           // 0 none, 1 QPSK, 2 8PSK
    int n_syms;
    int ra0, ra1, ra2;  // 8PSK column bases (d_rowaddr0..2)
    float rr, ri;       // (complexf) exp(-j pi/8)
};

} // namespace dvbs2
