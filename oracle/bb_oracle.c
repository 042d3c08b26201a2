/* oracle/bb_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked or called by the product).
 *
 * CPU restatement of the reference's baseband descrambler, the block that follows bch_decoder_bb in the
 * flowgraph (apps/dvbs2-rx:863-864):
 *   sequence   init_bb_derandomiser()        lib/bbdescrambler_bb_impl.cc:51-65
 *   work       out = in ^ sequence per frame lib/bbdescrambler_bb_impl.cc:67-82
 * The reference has no test for this block. The sequence is the DVB energy-dispersal PRBS (1 + x^14 + x^15,
 * register loaded with 100101010000000, ETSI EN 302 307-1 clause 5.2.2 = EN 300 421 clause 4.4.1), whose first
 * bytes 03 F6 08 34 30 B8 A3 93 are a published known answer; tests/test_oracle_kat.py pins the restatement to it.
 */
#include <stdint.h>
#include <string.h>

void oracle_bb_sequence(uint8_t* seq, int n_bytes)
{
    /* 15-bit register loaded with 100101010000000 (0x4A80 with the first stage in bit 0); output = feedback =
     * stage 14 xor stage 15 = bits 1 and 0 here; packed MSB first (lib/bbdescrambler_bb_impl.cc:51-65) */
    uint32_t reg = 0x4A80u;
    for (int byte = 0; byte < n_bytes; byte++) {
        uint32_t acc = 0;
        for (int bit = 0; bit < 8; bit++) {
            const uint32_t fb = (reg ^ (reg >> 1)) & 1u;
            acc = (acc << 1) | fb;
            reg = (reg >> 1) | (fb << 14);
        }
        seq[byte] = (uint8_t)acc;
    }
}

/* in/out: n_frames * kbch_bytes */
void oracle_bb_descramble(const uint8_t* in, uint8_t* out, int kbch_bytes, int n_frames)
{
    uint8_t seq[8100];
    oracle_bb_sequence(seq, kbch_bytes);
    for (int f = 0; f < n_frames; f++)
        for (int j = 0; j < kbch_bytes; j++) out[(size_t)f * kbch_bytes + j] = in[(size_t)f * kbch_bytes + j] ^ seq[j];
}
