/*
 * oracle/ldpc_oracle.c -- TEST INFRASTRUCTURE ONLY. Never linked into, imported by or called from
 * the product (gr-dvbs2rx_amd/); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may use it.
 *
 * Plain-C restatement of the reference's layered offset-min-sum int8 LDPC decoder for a group of
 * G frames that share one iteration count (the reference's SIMD batch):
 *
 *   LDPCDecoder<SIMD<int8_t,W>, OffsetMinSumAlgorithm<..., NormalUpdate, FACTOR=2>>
 *     init()        lib/ldpc_decoder/layered_decoder.hh:101-142  (pos/cnc construction, row permutation)
 *     bad()         lib/ldpc_decoder/layered_decoder.hh:32-49
 *     update()      lib/ldpc_decoder/layered_decoder.hh:50-79
 *     operator()    lib/ldpc_decoder/layered_decoder.hh:143-160
 *     finalp() etc. lib/ldpc_decoder/algorithms.hh:151-207
 *     lane ops      lib/ldpc_decoder/simd.hh:287-294 (vqabs), :1011-1019 (vqadd), :1085-1113 (vqsub),
 *                   :1142-1149 (vsign)
 *     table walk    lib/ldpc_decoder/ldpc.hh:44-87 (bit m of a group with row x -> checks (x+m*q) mod R)
 *
 * Parity pin: checked byte-for-byte (LLRs and return value) against the genuine reference decoders
 * built by oracle/Makefile into oracle/_ref/ (tests/test_oracle_kat.py::test_ldpc_oracle_vs_reference_live, wherever the prebuilt
 * oracle/_ref/ is present) and against the golden digests in tests/golden/ldpc_golden.json generated from those reference builds
 * (tools/gen_ldpc_golden.py; test_ldpc_oracle_matches_reference_digests).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct { const char* name; int N, K, nrows, off, nwords; } LdpcTableDesc;
#include "../gr-dvbs2rx_amd/csrc/ldpc_table_data.inc"

typedef struct {
    int N, K, M, R, q, CNL, LT;
    uint16_t* pos; /* [R][CNL], rows in layer order (layer i, lane j) = original check q*j+i */
    uint8_t* cnc;  /* [R] data degree per ORIGINAL check index (read with the layer index, like the reference) */
} Code;

static const LdpcTableDesc* find_table(const char* name)
{
    for (int i = 0; i < kNumLdpcTables; i++)
        if (!strcmp(kLdpcTableDescs[i].name, name)) return &kLdpcTableDescs[i];
    return 0;
}

/* layered_decoder.hh:101-142 + ldpc.hh:44-87 */
static int code_init(Code* c, const char* name)
{
    const LdpcTableDesc* t = find_table(name);
    if (!t) return -1;
    c->N = t->N; c->K = t->K; c->M = 360; c->R = c->N - c->K; c->q = c->R / c->M;
    const uint16_t* w = kLdpcTableWords + t->off;
    /* LINKS_MAX_CN: maximum check degree (data + 2 parity links) */
    int* cnt = (int*)calloc(c->R, sizeof(int));
    long links = 0;
    const uint16_t* p = w;
    for (int g = 0; g < t->nrows; g++) {
        int deg = *p++;
        for (int m = 0; m < c->M; m++)
            for (int n = 0; n < deg; n++) cnt[(p[n] + m * c->q) % c->R]++;
        p += deg; links += 360L * deg;
    }
    int maxc = 0;
    for (int i = 0; i < c->R; i++) if (cnt[i] > maxc) maxc = cnt[i];
    c->CNL = maxc;               /* = LINKS_MAX_CN - 2 */
    c->LT = (int)(links + 2L * c->R - 1); /* = LINKS_TOTAL */
    uint16_t* pos = (uint16_t*)calloc((size_t)c->R * c->CNL, sizeof(uint16_t));
    c->cnc = (uint8_t*)calloc(c->R, 1);
    p = w;
    int j = 0;
    for (int g = 0; g < t->nrows; g++) {
        int deg = *p++;
        for (int m = 0; m < c->M; m++, j++)
            for (int n = 0; n < deg; n++) {
                int i = (p[n] + m * c->q) % c->R;
                pos[c->CNL * i + c->cnc[i]++] = (uint16_t)j;
            }
        p += deg;
    }
    c->pos = (uint16_t*)calloc((size_t)c->R * c->CNL, sizeof(uint16_t));
    for (int i = 0; i < c->q; i++)
        for (int jj = 0; jj < c->M; jj++)
            for (int k = 0; k < c->CNL; k++)
                c->pos[c->CNL * (c->M * i + jj) + k] = pos[c->CNL * (c->q * jj + i) + k];
    free(pos); free(cnt);
    return 0;
}
static void code_free(Code* c) { free(c->pos); free(c->cnc); }

/* ---- lane arithmetic (simd.hh generic lanes; avx2.hh is bit-identical) ---- */
static inline int8_t qadd(int8_t a, int8_t b) { int x = a + b; return (int8_t)(x < -128 ? -128 : x > 127 ? 127 : x); }
static inline int8_t qsub(int8_t a, int8_t b) { int x = a - b; return (int8_t)(x < -128 ? -128 : x > 127 ? 127 : x); }
static inline int8_t qabs(int8_t a) { int x = a < -127 ? -127 : a; return (int8_t)(x < 0 ? -x : x); }
static inline int8_t vsign(int8_t a, int8_t b) { return (int8_t)(b > 0 ? a : b < 0 ? -a : 0); }
static inline int8_t imin(int8_t a, int8_t b) { return a < b ? a : b; }
static inline int8_t imax(int8_t a, int8_t b) { return a > b ? a : b; }

#define MAXDEG 64

/* algorithms.hh:170-192 for one lane */
static void finalp(int8_t* links, int cnt)
{
    int8_t mags[MAXDEG];
    for (int i = 0; i < cnt; i++) {
        uint8_t a = (uint8_t)qabs(links[i]);
        mags[i] = (int8_t)(a > 1 ? a - 1 : 0); /* unsigned saturating subtract of beta = 1 */
    }
    int8_t m0 = imin(mags[0], mags[1]), m1 = imax(mags[0], mags[1]);
    for (int i = 2; i < cnt; i++) {
        m1 = imin(m1, imax(m0, mags[i]));
        m0 = imin(m0, mags[i]);
    }
    int8_t signs = links[0];
    for (int i = 1; i < cnt; i++) signs ^= links[i];
    for (int i = 0; i < cnt; i++) {
        int8_t other = (mags[i] == m0) ? m1 : m0;
        links[i] = vsign(other, (int8_t)((signs ^ links[i]) | 127));
    }
}

/* layered_decoder.hh:32-49 ; data/parity are [bit][G] */
static int bad(const Code* c, const int8_t* data, const int8_t* pty, int G)
{
    const int M = c->M, q = c->q, CNL = c->CNL;
    for (int i = 0; i < q; i++) {
        int cnt = c->cnc[i];
        for (int j = 0; j < M; j++) {
            for (int l = 0; l < G; l++) {
                int8_t cnv = vsign(1, pty[(M * i + j) * G + l]);
                if (i) cnv = vsign(cnv, pty[(M * (i - 1) + j) * G + l]);
                else if (j) cnv = vsign(cnv, pty[(j + (q - 1) * M - 1) * G + l]);
                for (int k = 0; k < cnt; k++)
                    cnv = vsign(cnv, data[c->pos[CNL * (M * i + j) + k] * G + l]);
                if (cnv <= 0) return 1;
            }
        }
    }
    return 0;
}

/* layered_decoder.hh:50-79 */
static void update(const Code* c, int8_t* data, int8_t* pty, int8_t* bnl, int G)
{
    const int M = c->M, q = c->q, CNL = c->CNL;
    int8_t* bl = bnl;
    int8_t inp[MAXDEG], out[MAXDEG];
    for (int i = 0; i < q; i++) {
        int cnt = c->cnc[i];
        for (int j = 0; j < M; j++) {
            int deg = cnt + 2 - !(i | j);
            const uint16_t* ps = c->pos + CNL * (M * i + j);
            int pprev = i ? M * (i - 1) + j : j + (q - 1) * M - 1;
            for (int l = 0; l < G; l++) {
                for (int k = 0; k < cnt; k++) inp[k] = out[k] = qsub(data[ps[k] * G + l], bl[k * G + l]);
                inp[cnt] = out[cnt] = qsub(pty[(M * i + j) * G + l], bl[cnt * G + l]);
                if (i | j) inp[cnt + 1] = out[cnt + 1] = qsub(pty[pprev * G + l], bl[(cnt + 1) * G + l]);
                finalp(out, deg);
                for (int k = 0; k < cnt; k++) data[ps[k] * G + l] = qadd(inp[k], out[k]);
                pty[(M * i + j) * G + l] = qadd(inp[cnt], out[cnt]);
                if (i | j) pty[pprev * G + l] = qadd(inp[cnt + 1], out[cnt + 1]);
                for (int d = 0; d < deg; d++) /* algorithms.hh:203-206, NormalUpdate generic.hh:19-22 */
                    bl[d * G + l] = imin(imax(out[d], -32), 31);
            }
            bl += (size_t)deg * G;
        }
    }
}

/* layered_decoder.hh:143-160. code: G frames x N int8, frame-major, decoded in place.
 * Returns trials remaining (>= 0) or -1, exactly like ldpc_dec_decode(). */
int oracle_ldpc_decode(const char* table, int G, int8_t* code, int trials)
{
    Code c;
    if (code_init(&c, table)) return -1000;
    const int N = c.N, K = c.K, M = c.M, q = c.q;
    int8_t* data = (int8_t*)malloc((size_t)N * G);
    int8_t* pty = (int8_t*)malloc((size_t)c.R * G);
    int8_t* bnl = (int8_t*)calloc((size_t)c.LT * G, 1);
    for (int n = 0; n < G; n++)
        for (int j = 0; j < N; j++) data[j * G + n] = code[(size_t)n * N + j];
    int8_t* parity = data + (size_t)K * G;
    for (int i = 0; i < q; i++)
        for (int j = 0; j < M; j++) memcpy(pty + (size_t)(M * i + j) * G, parity + (size_t)(q * j + i) * G, G);
    while (bad(&c, data, pty, G) && --trials >= 0) update(&c, data, pty, bnl, G);
    for (int i = 0; i < q; i++)
        for (int j = 0; j < M; j++) memcpy(parity + (size_t)(q * j + i) * G, pty + (size_t)(M * i + j) * G, G);
    for (int n = 0; n < G; n++)
        for (int j = 0; j < N; j++) code[(size_t)n * N + j] = data[j * G + n];
    free(data); free(pty); free(bnl); code_free(&c);
    return trials;
}

/* Hard decision + MSB-first bit packing of ldpc_decoder_bb_impl::general_work
 * (lib/ldpc_decoder_bb_impl.cc:432-442). out_bytes per frame = nbits/8. */
void oracle_ldpc_pack(const int8_t* code, int n_frames, int N, int nbits, uint8_t* out)
{
    for (int f = 0; f < n_frames; f++)
        for (int j = 0; j < nbits / 8; j++) {
            uint8_t b = 0;
            for (int k = 0; k < 8; k++)
                if (code[(size_t)f * N + j * 8 + k] < 0) b |= (uint8_t)(1 << (7 - k));
            out[(size_t)f * (nbits / 8) + j] = b;
        }
}

/* Systematic IRA encoder over the same address tables (test-data generator; the reference has no
 * encoder -- its Tx side is gr-dtv). p[(x + m*q) mod R] ^= info bit, then p[r] ^= p[r-1]. */
int oracle_ldpc_encode(const char* table, const uint8_t* info /*K bits, one per byte*/, uint8_t* cw /*N*/)
{
    const LdpcTableDesc* t = find_table(table);
    if (!t) return -1;
    int N = t->N, K = t->K, R = N - K, q = R / 360;
    memcpy(cw, info, K);
    memset(cw + K, 0, R);
    const uint16_t* p = kLdpcTableWords + t->off;
    int j = 0;
    for (int g = 0; g < t->nrows; g++) {
        int deg = *p++;
        for (int m = 0; m < 360; m++, j++)
            if (info[j])
                for (int n = 0; n < deg; n++) cw[K + (p[n] + m * q) % R] ^= 1;
        p += deg;
    }
    for (int r = 1; r < R; r++) cw[K + r] ^= cw[K + r - 1];
    return 0;
}

int oracle_ldpc_info(const char* table, int* N, int* K, int* q, int* LT)
{
    Code c;
    if (code_init(&c, table)) return -1;
    *N = c.N; *K = c.K; *q = c.q; *LT = c.LT;
    code_free(&c);
    return 0;
}
