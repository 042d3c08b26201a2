/* oracle/pl_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked or called by the product).
 *
 * CPU restatement of the PLFRAME payload step of the reference's PL synchroniser, the stage that
 * hands XFECFRAME symbols to xfecframe_demapper_cb:
 *   descrambling sequence  pl_descrambler::compute_descrambling_sequence()   lib/pl_descrambler.cc:36-99
 *   descramble             pl_descrambler::descramble()                      lib/pl_descrambler.cc:101-105
 *   pilot removal + phase de-rotation per 16-slot segment  plsync_cc_impl::handle_payload()  lib/plsync_cc_impl.cc:644-653,
 *                                                          :727-795; index arithmetic :480-485; sizes lib/pl_signaling.cc:51-60
 * The reference has no unit test for this step (only GNU Radio flowgraph tests, python/dvbs2rx/qa_plsync_cc.py).
 * The Rn sequence is pinned by construction against the textbook definition of ETSI EN 302 307-1 clause 5.5.4
 * (tests/test_oracle_kat.py: the product computes it from the definition, this file by the reference's register
 * masks). The multiply and the rotator live in VOLK (volk_32fc_x2_multiply_32fc, volk_32fc_s32fc_x2_rotator_32fc, not
 * in the reference tree): restated from VOLK's published generic kernels (phase *= phase_inc per sample,
 * renormalised every 512 samples) -- PARITY UNPINNED for the float rounding, compared with a tolerance.
 */
#include <math.h>
#include <stdint.h>

/* 18-bit Fibonacci registers, bit 0 = oldest element of the m-sequence; parity of the masked register = an XOR of
 * future sequence elements (the shift-and-add property the reference exploits, lib/pl_descrambler.cc:23-34). */
static uint32_t tap(uint32_t reg, uint32_t mask) { return (uint32_t)__builtin_parity(reg & mask & 0x3FFFFu); }
static uint32_t advance(uint32_t reg, uint32_t feedback_mask) { return (reg >> 1) | (tap(reg, feedback_mask) << 17); }

/* Rn[i] in 0..3 for i < n. Feedback x: x^18 + x^7 + 1 (mask 0x81); y: y^18 + y^10 + y^7 + y^5 + 1 (mask 0x4A1);
 * the element 131072 positions ahead is tap 0x8050 of x and tap 0xFF60 of y (lib/pl_descrambler.cc:62-98). */
void oracle_pl_rn(int gold_code, uint8_t* rn, int n)
{
    uint32_t xr = 1u, yr = 0x3FFFFu;
    while (gold_code-- > 0) xr = advance(xr, 0x0081u);
    for (int i = 0; i < n; i++, xr = advance(xr, 0x0081u), yr = advance(yr, 0x04A1u)) {
        const uint32_t z_now = (xr ^ yr) & 1u;
        const uint32_t z_far = tap(xr, 0x8050u) ^ tap(yr, 0xFF60u);
        rn[i] = (uint8_t)(2u * z_far + z_now);
    }
}

/* one PLFRAME payload (n_slots*90 + n_pilots*36 symbols, interleaved re/im) -> n_slots*90 XFECFRAME symbols */
void oracle_pl_payload(const float* in, int n_slots, int has_pilots, int gold_code, float plheader_phase,
                       float fine_foffset, int coarse_corrected, const float* pilot_phase, float* out)
{
    static const float lut[4][2] = { { 1, 0 }, { 0, -1 }, { -1, 0 }, { 0, 1 } };
    const int n_pilots = has_pilots ? ((n_slots - 1) >> 4) : 0;
    const int payload_len = n_slots * 90 + n_pilots * 36;
    static uint8_t rn[360 * 90 + 22 * 36];
    oracle_pl_rn(gold_code, rn, payload_len);
    const float phase_inc = coarse_corrected ? (float)(2.0 * M_PI * fine_foffset) : 0.0f;
    const float ir = cosf(-phase_inc), ii = sinf(-phase_inc);
    float pr = cosf(-plheader_phase), pi = sinf(-plheader_phase);
    int produced = 0, counter = 0;
    for (int slot = 0; slot < n_slots; slot++) {
        const int blk = has_pilots ? slot / 16 : 0;
        if (has_pilots && slot % 16 == 0) {
            counter = 0; /* the reference makes one rotator call per 16-slot segment (:742-777) */
            if (coarse_corrected && blk > 0) { pr = cosf(-pilot_phase[blk - 1]); pi = sinf(-pilot_phase[blk - 1]); } /* :759-763 */
        }
        for (int s = 0; s < 90; s++) {
            const int k = slot * 90 + blk * 36 + s;
            const float xr = in[2 * k], xi = in[2 * k + 1];
            const float dr = xr * lut[rn[k]][0] - xi * lut[rn[k]][1], di = xr * lut[rn[k]][1] + xi * lut[rn[k]][0];
            out[2 * produced] = dr * pr - di * pi;
            out[2 * produced + 1] = dr * pi + di * pr;
            const float nr = pr * ir - pi * ii, ni = pr * ii + pi * ir;
            pr = nr; pi = ni;
            if (++counter == 512) { const float m = sqrtf(pr * pr + pi * pi); pr /= m; pi /= m; counter = 0; }
            produced++;
        }
    }
}
