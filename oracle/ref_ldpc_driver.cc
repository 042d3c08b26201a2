// oracle/ref_ldpc_driver.cc -- TEST INFRASTRUCTURE ONLY; BUILD CONTAINER ONLY.
//
// Thin C entry points around the GENUINE reference LDPC decoders. Nothing of the reference is
// copied: oracle/Makefile compiles /root/reference/lib/ldpc_decoder/ldpc_decoder_{avx2,sse41,generic}.cc
// where they lie and this driver only includes the reference's own headers by -I path
// (ldpc.hh, dvb_s2_tables.hh, dvb_s2x_tables.hh, dvb_t2_tables.hh). The result goes to oracle/_ref/
// (git-ignored, travels to the GPU box as a prebuilt .so).
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include "ldpc.hh"
#include "dvb_s2_tables.hh"
#include "dvb_s2x_tables.hh"
#include "dvb_t2_tables.hh"

#define NS(ns) namespace ns { void ldpc_dec_init(LDPCInterface* it); int ldpc_dec_decode(void* buffer, int8_t* code, int trials); }
NS(ldpc_avx2) NS(ldpc_sse41) NS(ldpc_generic)

static LDPCInterface* make_table(const char* n)
{
#define T(x) if (!std::strcmp(n, #x)) return new LDPC<DVB_##x>();
    T(S2_TABLE_B1) T(S2_TABLE_B2) T(S2_TABLE_B3) T(S2_TABLE_B4) T(S2_TABLE_B5) T(S2_TABLE_B6) T(S2_TABLE_B7)
    T(S2_TABLE_B8) T(S2_TABLE_B9) T(S2_TABLE_B10) T(S2_TABLE_B11) T(S2_TABLE_C1) T(S2_TABLE_C2) T(S2_TABLE_C3)
    T(S2_TABLE_C4) T(S2_TABLE_C5) T(S2_TABLE_C6) T(S2_TABLE_C7) T(S2_TABLE_C8) T(S2_TABLE_C9) T(S2_TABLE_C10)
    T(S2X_TABLE_B1) T(S2X_TABLE_B2) T(S2X_TABLE_B3) T(S2X_TABLE_B4) T(S2X_TABLE_B5) T(S2X_TABLE_B6)
    T(S2X_TABLE_B7) T(S2X_TABLE_B8) T(S2X_TABLE_B9) T(S2X_TABLE_B10) T(S2X_TABLE_B11) T(S2X_TABLE_B12)
    T(S2X_TABLE_B13) T(S2X_TABLE_B14) T(S2X_TABLE_B15) T(S2X_TABLE_B16) T(S2X_TABLE_B17) T(S2X_TABLE_B18)
    T(S2X_TABLE_B19) T(S2X_TABLE_B20) T(S2X_TABLE_B21) T(S2X_TABLE_B22) T(S2X_TABLE_B23) T(S2X_TABLE_B24)
    T(S2X_TABLE_C1) T(S2X_TABLE_C2) T(S2X_TABLE_C3) T(S2X_TABLE_C4) T(S2X_TABLE_C5) T(S2X_TABLE_C6)
    T(S2X_TABLE_C7) T(S2X_TABLE_C8) T(S2X_TABLE_C9) T(S2X_TABLE_C10) T(T2_TABLE_A3) T(T2_TABLE_B3)
#undef T
    return nullptr;
}

static int g_impl = 0, g_N = 0;
static void* g_buf = nullptr;

// impl: 0 = avx2 (32 frames/batch), 1 = sse4.1 (16), 2 = generic (16). Returns frames per batch or <0.
extern "C" int ref_ldpc_init(const char* table, int impl)
{
    LDPCInterface* t = make_table(table);
    if (!t) return -1;
    g_impl = impl; g_N = t->code_len();
    int simd = impl == 0 ? 32 : 16;
    if (impl == 0) ldpc_avx2::ldpc_dec_init(t);
    else if (impl == 1) ldpc_sse41::ldpc_dec_init(t);
    else ldpc_generic::ldpc_dec_init(t);
    delete t;
    std::free(g_buf);
    g_buf = aligned_alloc(32, (size_t)simd * g_N);
    return simd;
}

// code: simd*N int8, frame-major, decoded in place (lib/ldpc_decoder_bb_impl.cc:410). Returns the
// reference's return value (trials remaining, or -1).
extern "C" int ref_ldpc_decode(int8_t* code, int trials)
{
    if (g_impl == 0) return ldpc_avx2::ldpc_dec_decode(g_buf, code, trials);
    if (g_impl == 1) return ldpc_sse41::ldpc_dec_decode(g_buf, code, trials);
    return ldpc_generic::ldpc_dec_decode(g_buf, code, trials);
}
