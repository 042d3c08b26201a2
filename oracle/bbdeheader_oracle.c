/* oracle/bbdeheader_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked or called by the product).
 *
 * CPU restatement of the reference's BBFRAME de-header block, the block that follows bbdescrambler_bb in the flowgraph:
 *   CRC-8          bbdeheader_bb_impl::check_crc8        lib/bbdeheader_bb_impl.cc:138-142 (gf2_poly_rem, lib/gf_util.h:219-262,
 *                  generator x^8 + x^7 + x^6 + x^4 + x^2 + 1, lib/bbdeheader_bb_impl.cc:55)
 *   BBHEADER       bbdeheader_bb_impl::parse_bbheader    lib/bbdeheader_bb_impl.cc:77-136
 *   TS extraction  bbdeheader_bb_impl::general_work      lib/bbdeheader_bb_impl.cc:144-264
 * PINNED BY: the CRC against the genuine gf2_poly_rem (oracle/_ref, ref_crc8_rem in ref_bch_driver.cc); the header rules and
 * the packet state machine against the known answers of the reference's own test file python/dvbs2rx/qa_bbdeheader_bb.py
 * (its seven cases are restated as generators + expected user-packet ranges in tests/test_bbdeheader.py).
 *
 * ONE DEFINED DEVIATION: while re-synchronizing the reference subtracts SYNCD/8 + 1 from the unsigned count of DATAFIELD
 * bytes (:201-202); with SYNCD/8 + 1 > DFL/8 (a header that passes every check of parse_bbheader, e.g. SYNCD = DFL) the count
 * wraps and the packet loop reads past the input buffer -- undefined behaviour, no output to mirror. Here (and in the HIP
 * path) such a frame is dropped: counted in `overruns`, no bytes consumed from it, synched = false, partial count = 0.
 */
#include <stdint.h>
#include <string.h>

#define BBH_BYTES 10
#define TS_LEN 188

typedef struct {
    int kbch_bytes, max_dfl; /* bits */
    int synched, partial;
    uint8_t partial_pkt[TS_LEN];
    uint64_t packets, errors, bbframes, dropped, gaps, overruns;
} bbdh_oracle_t;

/* remainder of the byte string (as a polynomial, first byte = highest powers) modulo the generator; 0 = check passes */
uint8_t oracle_crc8_rem(const uint8_t* in, int size)
{
    /* bit-serial long division: what gf2_poly_rem computes with its byte table */
    uint32_t reg = 0; /* 9-bit window */
    for (int i = 0; i < size; i++)
        for (int b = 7; b >= 0; b--) {
            reg = (reg << 1) | ((in[i] >> b) & 1u);
            if (reg & 0x100u) reg ^= 0x1D5u; /* 0b111010101 */
        }
    return (uint8_t)reg;
}

void oracle_bbdh_init(bbdh_oracle_t* s, int kbch_bits)
{
    memset(s, 0, sizeof(*s));
    s->kbch_bytes = kbch_bits / 8;      /* :60 */
    s->max_dfl = kbch_bits - 80;        /* :61 */
}

/* parse_bbheader: 1 = valid. dfl / syncd returned in bits. */
static int parse(const bbdh_oracle_t* s, const uint8_t* in, unsigned* dfl, unsigned* syncd)
{
    if (oracle_crc8_rem(in, BBH_BYTES) != 0) return 0;                 /* :80-83 */
    const unsigned upl = ((unsigned)in[2] << 8) | in[3];               /* :100 */
    *dfl = ((unsigned)in[4] << 8) | in[5];                             /* :103 */
    *syncd = ((unsigned)in[7] << 8) | in[8];                           /* :108 */
    if (*dfl > (unsigned)s->max_dfl) return 0;                         /* :111 */
    if (*dfl % 8 != 0) return 0;                                       /* :116 */
    if (*syncd > *dfl) return 0;                                       /* :121 */
    if (upl != TS_LEN * 8) return 0;                                   /* :126 */
    if (*syncd % 8 != 0) return 0;                                     /* :131 */
    return 1;
}

/* general_work over n_frames whole BBFRAMEs; out must hold n_frames * ((max_dfl/8 + 187) / 188) * 188 bytes. Returns bytes produced. */
long long oracle_bbdh_work(bbdh_oracle_t* s, const uint8_t* in, int n_frames, uint8_t* out)
{
    long long produced = 0;
    for (int f = 0; f < n_frames; f++) {
        const uint8_t* p = in + (size_t)f * s->kbch_bytes;
        unsigned dfl, syncd;
        const int valid = parse(s, p, &dfl, &syncd);
        s->bbframes++;                                                  /* :164 */
        if (!valid) { s->synched = 0; s->dropped++; continue; }         /* :165-170 */
        p += BBH_BYTES;
        unsigned rem = dfl / 8;                                         /* :190 */
        if (s->partial > 0 && (int)(syncd / 8) != TS_LEN - 1 - s->partial) { s->synched = 0; s->gaps++; } /* :194-199 */
        if (!s->synched) {                                              /* :203-209 */
            const unsigned skip = syncd / 8 + 1;
            if (skip > rem) { s->overruns++; s->synched = 0; s->partial = 0; continue; } /* the defined deviation (header) */
            p += skip; rem -= skip;
            s->synched = 1; s->partial = 0;
        }
        while (rem >= TS_LEN) {                                         /* :212-238 */
            const uint8_t* pkt;
            if (s->partial > 0) {
                const unsigned need = TS_LEN - (unsigned)s->partial;
                memcpy(s->partial_pkt + s->partial, p, need);
                s->partial = 0; p += need; rem -= need;
                pkt = s->partial_pkt;
            } else { pkt = p; p += TS_LEN; rem -= TS_LEN; }
            const int ok = oracle_crc8_rem(pkt, TS_LEN) == 0;
            out[0] = 0x47;
            memcpy(out + 1, pkt, TS_LEN - 1);
            if (!ok) { out[1] |= 0x80; s->errors++; }
            out += TS_LEN; produced += TS_LEN; s->packets++;
        }
        if (rem > 0) { s->partial = (int)rem; memcpy(s->partial_pkt, p, rem); } /* :241-245 */
    }
    return produced;
}
