// oracle/ref_bch_driver.cc -- TEST INFRASTRUCTURE ONLY; BUILD CONTAINER ONLY.
//
// Thin C entry points around the GENUINE reference BCH codec. Nothing of the reference is copied: oracle/Makefile
// compiles /root/reference/lib/bch.cc and lib/gf.cc where they lie, and this driver only includes the reference's own
// lib/bch.h. The one header of the reference that needs GNU Radio, include/gnuradio/dvbs2rx/api.h (it only defines
// the DVBS2RX_API export attribute from <gnuradio/attributes.h>), is switched off by its own include guard on the
// command line (-DINCLUDED_DVBS2RX_API_H -DDVBS2RX_API=): no stand-in header file is written, every line that is
// compiled is the reference's. The result goes to oracle/_ref/ (git-ignored, travels to the GPU box prebuilt).
#include <cstdint>
#include <exception>
#include "bch.h"
#include "gf_util.h"

using namespace gr::dvbs2rx;
typedef bch_codec<uint32_t, bitset256_t> codec_t; // the instantiation of bch_decoder_bb_impl (lib/bch_decoder_bb_impl.h)

struct ref_bch { galois_field<uint32_t>* gf; codec_t* codec; };

// prim_poly: bit i = coefficient of x^i, as the block passes it (lib/bch_decoder_bb_impl.cc:58-63). n = 0: 2^m - 1.
extern "C" void* ref_bch_new(uint32_t prim_poly, int t, int n)
{
    try {
        ref_bch* h = new ref_bch();
        h->gf = new galois_field<uint32_t>(prim_poly);
        h->codec = new codec_t(h->gf, (uint8_t)t, (uint32_t)n);
        return h;
    } catch (const std::exception&) { return nullptr; }
}
extern "C" void ref_bch_free(void* p)
{
    ref_bch* h = (ref_bch*)p;
    if (!h) return;
    delete h->codec; delete h->gf; delete h;
}
extern "C" int ref_bch_k(void* p) { return (int)((ref_bch*)p)->codec->get_k(); }
extern "C" int ref_bch_n(void* p) { return (int)((ref_bch*)p)->codec->get_n(); }
// bch_codec::decode(u8_cptr_t, u8_ptr_t) (lib/bch.cc:468-487). Returns its return value, or -2 when it throws
// (std::out_of_range from galois_field::get_exponent(0), lib/gf.h:110; std::runtime_error, lib/bch.cc:443-444).
extern "C" int ref_bch_decode(void* p, const uint8_t* cw, uint8_t* msg)
{
    try { return ((ref_bch*)p)->codec->decode(cw, msg); }
    catch (const std::exception&) { return -2; }
}
extern "C" void ref_bch_encode(void* p, const uint8_t* msg, uint8_t* cw) { ((ref_bch*)p)->codec->encode(msg, cw); }

// The CRC-8 check of bbdeheader_bb (lib/bbdeheader_bb_impl.cc:55-56,138-142): remainder of the byte string modulo
// x^8 + x^7 + x^6 + x^4 + x^2 + 1 through the reference's own table and gf2_poly_rem (lib/gf_util.h:177-262).
extern "C" int ref_crc8_rem(const uint8_t* in, int size)
{
    static const gf2_poly<uint16_t> poly(0b111010101);
    static const std::array<uint16_t, 256> lut = build_gf2_poly_rem_lut(poly);
    return (int)gf2_poly_rem(in, size, poly, lut).get_poly();
}
