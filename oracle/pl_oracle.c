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

static int par18(long a, long b) { a &= b; int c = 0; for (int i = 0; i < 18; i++) c += (a >> i) & 1; return c & 1; }

/* Rn[i] in 0..3 for i < n (lib/pl_descrambler.cc:62-98) */
void oracle_pl_rn(int gold_code, uint8_t* rn, int n)
{
    long x = 0x00001, y = 0x3FFFF;
    for (int k = 0; k < gold_code; k++) { int xb = par18(x, 0x0081); x >>= 1; if (xb) x |= 0x20000; }
    for (int i = 0; i < n; i++) {
        int xa = par18(x, 0x8050), xb = par18(x, 0x0081), xc = (int)(x & 1);
        x >>= 1; if (xb) x |= 0x20000;
        int ya = par18(y, 0x04A1), yb = par18(y, 0xFF60), yc = (int)(y & 1);
        y >>= 1; if (ya) y |= 0x20000;
        rn[i] = (uint8_t)(((xa ^ yb) << 1) + (xc ^ yc));
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
