/*
 * oracle/demap_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle/ldpc_oracle.c header for the rule).
 *
 * Plain-C restatement of the soft constellation demappers behind xfecframe_demapper_cb:
 *   QPSK  QpskConstellation::demap_soft            lib/qpsk.h:208-214
 *         (arithmetic lives in VOLK's volk_32f_s32f_convert_8i, a third-party dependency that is NOT
 *          in /root/reference and whose version the reference does not pin (GNU Radio >= 3.10,
 *          CMakeLists.txt:77). Restated from VOLK's published generic kernel:
 *          out = (int8) clamp(rintf(in * scalar), -128, 127), round-to-nearest-even.)
 *   8PSK  PhaseShiftKeying<8,gr_complex,int8_t>::soft  lib/psk.hh:143-150 (quantize :123-131, rot :113)
 *         + column de-interleave lib/xfecframe_demapper_cb_impl.cc:155-176 (row addresses :50-69)
 *   SNR   pre-decoder estimate, 8PSK loop         lib/xfecframe_demapper_cb_impl.cc:132-145
 *         (hard :135-141 / map :152-157 of lib/psk.hh); QPSK lib/qpsk.h:41-65,171-181,240-244
 *
 * PARITY UNPINNED for the rounding/saturation of both demappers: the only reference test at this
 * boundary is lib/qa_qpsk.cc:67-79 (exact-integer inputs, reproduced in tests/golden/demap_kat.json);
 * lib/psk.hh has no test and is not buildable here without a stand-in <gnuradio/gr_complex.h>.
 * Compile with -ffp-contract=off: the reference's `c *= rot` is FMA-contraction sensitive.
 */
#include <math.h>
#include <stdint.h>

static inline int8_t sat8(float v)
{
    if (v > 127.0f) return 127;
    if (v < -128.0f) return -128;
    return (int8_t)rintf(v);
}

/* lib/qpsk.h:208-214. syms: n_syms interleaved (re, im) floats; out: 2*n_syms int8 */
void oracle_demap_qpsk(const float* syms, int n_syms, float N0, int8_t* out)
{
    float scalar = (float)(2 * M_SQRT2 / N0);
    for (int i = 0; i < 2 * n_syms; i++) out[i] = sat8(syms[i] * scalar);
}

static inline int8_t quant8(float precision, float value) /* lib/psk.hh:123-131 */
{
    const float sin_pi_8 = 0.38268343236508977173f;
    const float DIST = 2 * sin_pi_8;
    value *= DIST * precision;
    value = nearbyintf(value);
    value = fminf(fmaxf(value, -128.0f), 127.0f);
    return (int8_t)value;
}

/* lib/psk.hh:143-150 for one symbol; b[3] */
static inline void psk8_soft(float re, float im, float precision, int8_t* b)
{
    const float rcp_sqrt_2 = 0.70710678118654752440f;
    const float rr = (float)cos(-M_PI / 8), ri = (float)sin(-M_PI / 8); /* (complexf) exp(-j pi/8) */
    float cr = re * rr - im * ri;
    float ci = re * ri + im * rr;
    b[1] = quant8(precision, cr);
    b[2] = quant8(precision, ci);
    b[0] = quant8(precision, rcp_sqrt_2 * (fabsf(cr) - fabsf(ci)));
}

/* column order: 0 -> "012", 1 -> "210" (C3_5), 2 -> "102" (C25_36, C13_18, C7_15, C8_15, C26_45) */
void oracle_demap_8psk(const float* syms, int n_syms, float N0, int order, int8_t* out)
{
    float precision = (float)(4.0 / N0);
    int rows = n_syms;
    int ra0 = 0, ra1 = rows, ra2 = 2 * rows;
    if (order == 1) { ra0 = 2 * rows; ra1 = rows; ra2 = 0; }
    else if (order == 2) { ra0 = rows; ra1 = 0; ra2 = 2 * rows; }
    for (int j = 0; j < n_syms; j++) {
        int8_t b[3];
        psk8_soft(syms[2 * j], syms[2 * j + 1], precision, b);
        out[ra0 + j] = b[0]; out[ra1 + j] = b[1]; out[ra2 + j] = b[2];
    }
}

/* pre-decoder SNR estimate of the block (linear). constellation: 4 or 8.
 * 8PSK: lib/xfecframe_demapper_cb_impl.cc:132-145 (sequential float accumulation).
 * QPSK: lib/qpsk.h:240-244 -> :171-181 (slice) -> :41-65 (VOLK dot products; reduction order is
 * implementation-defined there -- this restatement accumulates sequentially in float). */
float oracle_demap_snr(const float* syms, int n_syms, int constellation)
{
    const float rcp_sqrt_2 = 0.70710678118654752440f;
    float sp = 0, np = 0;
    if (constellation == 4) {
        for (int j = 0; j < n_syms; j++) {
            float sr = syms[2 * j] >= 0 ? rcp_sqrt_2 : -rcp_sqrt_2;
            float si = syms[2 * j + 1] >= 0 ? rcp_sqrt_2 : -rcp_sqrt_2;
            float er = syms[2 * j] - sr, ei = syms[2 * j + 1] - si;
            sp += sr * sr + si * si; np += er * er + ei * ei;
        }
    } else {
        static const float m8[8][2] = { { 0.70710678118654752440f, 0.70710678118654752440f }, { 1, 0 }, { -1, 0 },
            { -0.70710678118654752440f, -0.70710678118654752440f }, { 0, 1 },
            { 0.70710678118654752440f, -0.70710678118654752440f }, { -0.70710678118654752440f, 0.70710678118654752440f }, { 0, -1 } };
        const float rr = (float)cos(-M_PI / 8), ri = (float)sin(-M_PI / 8);
        for (int j = 0; j < n_syms; j++) {
            float re = syms[2 * j], im = syms[2 * j + 1];
            float cr = re * rr - im * ri, ci = re * ri + im * rr;
            int b1 = cr < 0 ? -1 : 1, b2 = ci < 0 ? -1 : 1, b0 = fabsf(cr) < fabsf(ci) ? -1 : 1;
            int idx = (((b0 + 1) << 1) ^ 0x4) | ((b1 + 1) ^ 0x2) | (((b2 + 1) >> 1) ^ 0x1);
            float er = re - m8[idx][0], ei = im - m8[idx][1];
            sp += m8[idx][0] * m8[idx][0] + m8[idx][1] * m8[idx][1]; np += er * er + ei * ei;
        }
    }
    if (!(np > 0)) np = 1e-12f;
    return sp / np;
}

/* post-decoder refinement for one frame: lib/xfecframe_demapper_cb_impl.cc:268-307 (8PSK: signs of the decoded
 * LLRs re-interleaved :274-291, mapped :297-302) and lib/qpsk.h:266-281 (QPSK: int8 -> float -> slice (>= 0 is
 * the positive point, :171-181) -> _estimate_snr :41-65). order = 8PSK column order as in oracle_demap_8psk. */
float oracle_demap_snr_refined(const float* syms, const int8_t* llr, int n_syms, int constellation, int order)
{
    const float rcp_sqrt_2 = 0.70710678118654752440f;
    float sp = 0, np = 0;
    if (constellation == 4) {
        for (int j = 0; j < n_syms; j++) {
            float sr = llr[2 * j] >= 0 ? rcp_sqrt_2 : -rcp_sqrt_2;
            float si = llr[2 * j + 1] >= 0 ? rcp_sqrt_2 : -rcp_sqrt_2;
            float er = syms[2 * j] - sr, ei = syms[2 * j + 1] - si;
            sp += sr * sr + si * si; np += er * er + ei * ei;
        }
    } else {
        static const float m8[8][2] = { { 0.70710678118654752440f, 0.70710678118654752440f }, { 1, 0 }, { -1, 0 },
            { -0.70710678118654752440f, -0.70710678118654752440f }, { 0, 1 },
            { 0.70710678118654752440f, -0.70710678118654752440f }, { -0.70710678118654752440f, 0.70710678118654752440f }, { 0, -1 } };
        int rows = n_syms, ra0 = 0, ra1 = rows, ra2 = 2 * rows;
        if (order == 1) { ra0 = 2 * rows; ra1 = rows; ra2 = 0; }
        else if (order == 2) { ra0 = rows; ra1 = 0; ra2 = 2 * rows; }
        for (int j = 0; j < n_syms; j++) {
            int b0 = llr[ra0 + j] < 0 ? -1 : 1, b1 = llr[ra1 + j] < 0 ? -1 : 1, b2 = llr[ra2 + j] < 0 ? -1 : 1;
            int idx = (((b0 + 1) << 1) ^ 0x4) | ((b1 + 1) ^ 0x2) | (((b2 + 1) >> 1) ^ 0x1);
            float er = syms[2 * j] - m8[idx][0], ei = syms[2 * j + 1] - m8[idx][1];
            sp += m8[idx][0] * m8[idx][0] + m8[idx][1] * m8[idx][1]; np += er * er + ei * ei;
        }
    }
    if (!(np > 0)) np = 1e-12f;
    return sp / np;
}
