// fec_tables.h -- static code tables and the (standard, framesize, rate) parameter map.
//
// Mirrors, as data + lookup code written for this repo:
//   get_fec_info()                      reference lib/fec_params.cc:16-344
//   rate/framesize -> LDPC table switch reference lib/ldpc_decoder_bb_impl.cc:104-307
//   DVB_S2[X]/T2_TABLE_* contents       reference lib/dvb_s2_tables.hh, dvb_s2x_tables.hh, dvb_t2_tables.hh
// Enum values follow reference include/gnuradio/dvbs2rx/dvb_config.h:15-116.
#pragma once
#include <cstdint>

namespace dvbs2 {

struct LdpcTableDesc {
    const char* name; // e.g. "S2_TABLE_B4"
    int N, K;         // code length, information length
    int nrows;        // K / 360 rows (one per 360-bit information group)
    int off, nwords;  // slice of the word stream: per row "deg, addr[0..deg)"
};

struct FecParamRow {
    int standard, framesize, rate;
    uint32_t bch_k, bch_n, bch_t, ldpc_n;
    const char* table;
};

struct FecInfo {
    uint32_t bch_k, bch_n, bch_t;
    uint32_t ldpc_k, ldpc_n; // ldpc_k == bch_n (reference lib/fec_params.cc:343)
    const LdpcTableDesc* table;
};

const LdpcTableDesc* find_ldpc_table(const char* name);
const uint16_t* ldpc_table_words(const LdpcTableDesc* t);
int num_ldpc_tables();
const LdpcTableDesc* ldpc_table_at(int i);

// Returns false when the reference's get_fec_info() leaves the triple unset.
bool get_fec_info(int standard, int framesize, int rate, FecInfo* out);
const char* rate_name(int rate);
int num_rates();

} // namespace dvbs2
