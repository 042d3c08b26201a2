/*
 * oracle/bch_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle/ldpc_oracle.c header for the rule).
 *
 * Plain-C restatement of the reference's GF(2^m) BCH codec as used by bch_decoder_bb:
 *   field tables            lib/gf.cc:20-66 (LFSR element table), :69-98 (mul/inv/div via exponents)
 *   minimal polynomials     lib/gf.cc:100-136 (conjugates, product of (x + beta^(2^l)))
 *   generator polynomial    lib/bch.cc:37-62  (product of DISTINCT minimal polys of alpha^1,3,..,2t-1)
 *   ctor / shortening       lib/bch.cc:64-113 (s = 2^m-1-n, k = n - deg g, quadratic LUT :107-112)
 *   encode (bytes)          lib/bch.cc:158-173
 *   syndrome                lib/bch.cc:176-189, :217-222 (remainder mod g, empty if zero, else r(alpha^i) i=1..2t)
 *   simplified Berlekamp    lib/bch.cc:225-304
 *   error-location numbers  lib/bch.cc:307-385 (+ Chien: lib/gf.cc:290-404)
 *   correction / decode     lib/bch.cc:429-452, :468-487
 *
 * Pinned three ways: the reference's own known-answer tests (lib/qa_bch.cc, lib/qa_gf.cc)
 * transcribed as data in tests/golden/bch_kat.json; digests of the GENUINE codec (oracle/_ref/libdvbs2_ref_bch.so =
 * lib/bch.cc + lib/gf.cc compiled where they lie, oracle/Makefile) in tests/golden/bch_golden.json, including words
 * crafted to reach both throw sites; and live against that library (tests/test_oracle_kat.py).
 *
 * Return convention of oracle_bch_decode_bytes: >=0 corrected count, -1 failure (as the reference),
 * -2 = the reference would have thrown (std::out_of_range from galois_field::get_exponent(0), lib/gf.h:110,
 * reached from err_loc_numbers' quadratic branch lib/bch.cc:359-367; or "Error location number out of
 * range", lib/bch.cc:443-444).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int m, t, n, k, s, nfield; /* nfield = 2^m - 1 */
    uint32_t* alpha;           /* alpha[i] = alpha^i, i in [0, nfield) */
    uint32_t* logt;            /* logt[x] = i with alpha^i = x (x != 0) */
    uint32_t* quad;            /* quadratic LUT: quad[r*r ^ r] = r */
    uint8_t* g;                /* generator polynomial coefficients g[i] of x^i */
    int gdeg;
} Bch;

static uint32_t gmul(const Bch* b, uint32_t x, uint32_t y)
{
    if (!x || !y) return 0;
    return b->alpha[(b->logt[x] + b->logt[y]) % b->nfield];
}
/* inverse(0) throws in the reference; callers must guard */
static uint32_t ginv(const Bch* b, uint32_t x) { return b->alpha[(b->nfield - b->logt[x]) % b->nfield]; }

Bch* oracle_bch_new(int m, uint32_t prim_poly, int t, int n)
{
    Bch* b = (Bch*)calloc(1, sizeof(Bch));
    b->m = m; b->t = t; b->nfield = (1 << m) - 1;
    b->alpha = (uint32_t*)calloc(b->nfield, 4);
    b->logt = (uint32_t*)calloc(b->nfield + 1, 4);
    uint32_t low = prim_poly ^ (1u << m), x = 1;
    for (int i = 0; i < b->nfield; i++) {
        b->alpha[i] = x; b->logt[x] = i;
        x = ((x << 1) & b->nfield) ^ ((x >> (m - 1)) * low);
    }
    /* generator polynomial: product of distinct minimal polynomials */
    uint8_t* seen = (uint8_t*)calloc(b->nfield + 1, 1);
    int cap = m * t + 2;
    uint8_t* g = (uint8_t*)calloc(cap, 1); g[0] = 1; int gdeg = 0;
    for (int i = 0; i < t; i++) {
        uint32_t e = (2 * i + 1) % b->nfield;
        if (seen[b->alpha[e]]) continue;
        /* conjugates alpha^(e*2^j) */
        uint32_t conj[32]; int nc = 0; uint32_t ee = e;
        for (int j = 0; j < m; j++) {
            uint32_t el = b->alpha[ee];
            int dup = 0;
            for (int c = 0; c < nc; c++) if (conj[c] == el) dup = 1;
            if (dup) break;
            conj[nc++] = el; seen[el] = 1;
            ee = (uint32_t)(((uint64_t)ee * 2) % b->nfield);
        }
        /* minimal polynomial = prod (x + conj) over GF(2^m); result is binary */
        uint32_t mp[40] = { 1 }; int md = 0;
        for (int c = 0; c < nc; c++) {
            uint32_t nx[40] = { 0 };
            for (int d = 0; d <= md; d++) { nx[d + 1] ^= mp[d]; nx[d] ^= gmul(b, mp[d], conj[c]); }
            md++; memcpy(mp, nx, sizeof(nx));
        }
        uint8_t* ng = (uint8_t*)calloc(cap + 40, 1);
        for (int a = 0; a <= gdeg; a++) if (g[a]) for (int d = 0; d <= md; d++) ng[a + d] ^= (uint8_t)(mp[d] & 1);
        gdeg += md; memcpy(g, ng, gdeg + 1); free(ng);
    }
    free(seen);
    b->g = g; b->gdeg = gdeg;
    b->n = n ? n : b->nfield; b->s = b->nfield - b->n; b->k = b->n - gdeg;
    b->quad = (uint32_t*)calloc(b->nfield + 1, 4);
    for (uint32_t r = 0; r <= (uint32_t)b->nfield; r++) b->quad[gmul(b, r, r) ^ r] = r;
    return b;
}
void oracle_bch_free(Bch* b) { free(b->alpha); free(b->logt); free(b->quad); free(b->g); free(b); }
int oracle_bch_k(const Bch* b) { return b->k; }
int oracle_bch_n(const Bch* b) { return b->n; }
int oracle_bch_gdeg(const Bch* b) { return b->gdeg; }
void oracle_bch_genpoly(const Bch* b, uint8_t* out) { memcpy(out, b->g, b->gdeg + 1); }
uint32_t oracle_bch_alpha(const Bch* b, int i) { return b->alpha[((i % b->nfield) + b->nfield) % b->nfield]; }
/* minimal polynomial of alpha^e as an integer bit mask (lib/gf.cc:118-136), for the qa_gf KATs */
uint32_t oracle_bch_minpoly(const Bch* b, int e)
{
    uint32_t conj[32]; int nc = 0; uint32_t ee = (uint32_t)(e % b->nfield);
    for (int j = 0; j < b->m; j++) {
        uint32_t el = b->alpha[ee]; int dup = 0;
        for (int c = 0; c < nc; c++) if (conj[c] == el) dup = 1;
        if (dup) break;
        conj[nc++] = el; ee = (uint32_t)(((uint64_t)ee * 2) % b->nfield);
    }
    uint32_t mp[40] = { 1 }; int md = 0;
    for (int c = 0; c < nc; c++) {
        uint32_t nx[40] = { 0 };
        for (int d = 0; d <= md; d++) { nx[d + 1] ^= mp[d]; nx[d] ^= gmul(b, mp[d], conj[c]); }
        md++; memcpy(mp, nx, sizeof(nx));
    }
    uint32_t r = 0;
    for (int d = 0; d <= md; d++) if (mp[d] & 1) r |= 1u << d;
    return r;
}

/* remainder of c(x) mod g(x); bits[0] = coefficient of x^(nbits-1). rem[i] = coef of x^i, i < gdeg */
static void poly_rem(const Bch* b, const uint8_t* bits, int nbits, uint8_t* rem)
{
    int gd = b->gdeg;
    memset(rem, 0, gd);
    for (int i = 0; i < nbits; i++) {
        uint8_t fb = rem[gd - 1];
        memmove(rem + 1, rem, gd - 1);
        rem[0] = bits[i];
        if (fb) for (int d = 0; d < gd; d++) rem[d] ^= b->g[d];
    }
}

/* bits API: msg k bits (msg[0] = highest order), cw n bits */
void oracle_bch_encode_bits(const Bch* b, const uint8_t* msg, uint8_t* cw)
{
    memcpy(cw, msg, b->k); memset(cw + b->k, 0, b->gdeg);
    uint8_t* rem = (uint8_t*)malloc(b->gdeg);
    poly_rem(b, cw, b->n, rem);
    for (int i = 0; i < b->gdeg; i++) cw[b->n - 1 - i] = rem[i];
    free(rem);
}

/* returns 0 when the remainder is zero (reference: empty vector), else 2t and fills S[0..2t) */
int oracle_bch_syndrome_bits(const Bch* b, const uint8_t* cw, uint32_t* S)
{
    uint8_t* rem = (uint8_t*)malloc(b->gdeg);
    poly_rem(b, cw, b->n, rem);
    int nz = 0;
    for (int i = 0; i < b->gdeg; i++) nz |= rem[i];
    if (!nz) { free(rem); return 0; }
    for (int i = 1; i <= 2 * b->t; i++) {
        uint32_t v = 0;
        for (int d = 0; d < b->gdeg; d++) if (rem[d]) v ^= b->alpha[(uint32_t)(((uint64_t)i * d) % b->nfield)];
        S[i - 1] = v;
    }
    free(rem);
    return 2 * b->t;
}

#define MAXT 32
typedef struct { uint32_t c[2 * MAXT + 4]; int deg; } Poly; /* deg = -1 for the zero polynomial */
static void ptrim(Poly* p, int top) { p->deg = top; while (p->deg >= 0 && p->c[p->deg] == 0) p->deg--; }

/* lib/bch.cc:225-304 */
int oracle_bch_err_loc_poly(const Bch* b, const uint32_t* S, uint32_t* sigma_out)
{
    int t = b->t, nrows = t + 2;
    int two_mu[MAXT + 2]; two_mu[0] = -1;
    for (int i = 0; i < t + 1; i++) two_mu[i + 1] = 2 * i;
    Poly sv[MAXT + 3]; memset(sv, 0, sizeof(sv));
    sv[0].c[0] = 1; sv[0].deg = 0;
    sv[1].c[0] = 1; sv[1].deg = 0;
    sv[2].c[0] = 1; sv[2].c[1] = S[0]; ptrim(&sv[2], 1);
    uint32_t d[MAXT + 2] = { 0 };
    d[0] = 1; d[1] = S[0];
    int row = 2;
    while (row <= t) {
        int tm = two_mu[row];
        d[row] = S[tm];
        for (int j = 1; j <= sv[row].deg; j++)
            if (sv[row].c[j]) d[row] ^= gmul(b, sv[row].c[j], S[tm - j]);
        if (d[row] == 0) sv[row + 1] = sv[row];
        else {
            int row_rho = 0, max_diff = -2;
            for (int j = row - 1; j >= 0; j--)
                if (d[j] != 0) {
                    int diff = two_mu[j] - sv[j].deg;
                    if (diff > max_diff) { max_diff = diff; row_rho = j; }
                }
            int shift = tm - two_mu[row_rho];
            uint32_t coef = gmul(b, d[row], ginv(b, d[row_rho]));
            Poly r; memset(&r, 0, sizeof(r));
            int top = sv[row].deg;
            for (int j = 0; j <= sv[row].deg; j++) r.c[j] = sv[row].c[j];
            for (int j = 0; j <= sv[row_rho].deg; j++) r.c[j + shift] ^= gmul(b, coef, sv[row_rho].c[j]);
            if (sv[row_rho].deg + shift > top) top = sv[row_rho].deg + shift;
            ptrim(&r, top);
            sv[row + 1] = r;
        }
        row++;
    }
    (void)nrows;
    for (int j = 0; j <= sv[row].deg; j++) sigma_out[j] = sv[row].c[j];
    return sv[row].deg;
}

/* lib/bch.cc:307-385. Returns count, or -2 if the reference would throw (get_exponent(0)). */
int oracle_bch_err_loc_numbers(const Bch* b, const uint32_t* sigma, int deg, uint32_t* numbers)
{
    if (deg > b->t) return 0;
    if (deg == 1) { numbers[0] = gmul(b, sigma[1], ginv(b, sigma[0])); return 1; }
    if (deg == 2) {
        if (sigma[1] == 0 || sigma[0] == 0) return 0;
        uint32_t b_over_a = gmul(b, sigma[1], ginv(b, sigma[2]));
        uint32_t rr = gmul(b, gmul(b, sigma[0], sigma[2]), ginv(b, gmul(b, sigma[1], sigma[1])));
        uint32_t r = b->quad[rr];
        uint32_t x0 = gmul(b, r, b_over_a), x1 = gmul(b, b_over_a, r ^ 1);
        if (x0 == 0 || x1 == 0) return -2;
        numbers[0] = ginv(b, x0); numbers[1] = ginv(b, x1);
        return 2;
    }
    /* Chien search over exponents [s+1, n+s], early stop at deg roots (lib/gf.cc:376-401) */
    int nfound = 0;
    for (uint32_t i = (uint32_t)b->s + 1; i <= (uint32_t)(b->n + b->s); i++) {
        uint32_t res = 0;
        for (int j = 0; j <= deg; j++)
            if (sigma[j]) res ^= b->alpha[(b->logt[sigma[j]] + (uint64_t)i * j) % b->nfield];
        if (res == 0) {
            numbers[nfound] = b->alpha[(b->nfield - i % b->nfield) % b->nfield];
            if (++nfound == deg) break;
        }
    }
    return nfound;
}

/* lib/bch.cc:468-487 with lib/bch.cc:429-452 */
int oracle_bch_decode_bytes(const Bch* b, const uint8_t* cw, uint8_t* msg)
{
    int nb = b->n / 8, kb = b->k / 8;
    memcpy(msg, cw, kb);
    uint8_t* bits = (uint8_t*)malloc(b->n);
    for (int i = 0; i < nb; i++) for (int j = 0; j < 8; j++) bits[8 * i + j] = (cw[i] >> (7 - j)) & 1;
    uint32_t S[2 * MAXT];
    int ns = oracle_bch_syndrome_bits(b, bits, S);
    free(bits);
    if (!ns) return 0;
    uint32_t sigma[2 * MAXT + 4], numbers[MAXT + 2];
    int deg = oracle_bch_err_loc_poly(b, S, sigma);
    int cnt = oracle_bch_err_loc_numbers(b, sigma, deg, numbers);
    if (cnt == -2) return -2;
    for (int i = 0; i < cnt; i++) {
        uint32_t bit_idx = b->logt[numbers[i]];
        if (bit_idx >= (uint32_t)b->n) return -2;
        if (bit_idx < (uint32_t)(b->n - b->k)) continue;
        uint32_t net = b->n - 1 - bit_idx;
        msg[net / 8] ^= (uint8_t)(1u << (7 - (net % 8)));
    }
    return deg == cnt ? cnt : -1;
}

void oracle_bch_encode_bytes(const Bch* b, const uint8_t* msg, uint8_t* cw)
{
    uint8_t* mb = (uint8_t*)malloc(b->n); uint8_t* cb = (uint8_t*)malloc(b->n);
    for (int i = 0; i < b->k / 8; i++) for (int j = 0; j < 8; j++) mb[8 * i + j] = (msg[i] >> (7 - j)) & 1;
    oracle_bch_encode_bits(b, mb, cb);
    memset(cw, 0, b->n / 8);
    for (int i = 0; i < b->n; i++) cw[i / 8] |= (uint8_t)(cb[i] << (7 - (i % 8)));
    free(mb); free(cb);
}
