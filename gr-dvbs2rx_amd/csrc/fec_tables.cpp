// fec_tables.cpp -- see fec_tables.h.
#include "fec_tables.h"
#include <cstring>

namespace dvbs2 {

#include "ldpc_table_data.inc"
#include "fec_params_data.inc"

const LdpcTableDesc* find_ldpc_table(const char* name)
{
    if (!name) return nullptr;
    if (!std::strncmp(name, "DVB_", 4)) name += 4;
    for (int i = 0; i < kNumLdpcTables; i++)
        if (!std::strcmp(kLdpcTableDescs[i].name, name)) return &kLdpcTableDescs[i];
    return nullptr;
}
const uint16_t* ldpc_table_words(const LdpcTableDesc* t) { return kLdpcTableWords + t->off; }
int num_ldpc_tables() { return kNumLdpcTables; }
const LdpcTableDesc* ldpc_table_at(int i) { return (i >= 0 && i < kNumLdpcTables) ? &kLdpcTableDescs[i] : nullptr; }

bool get_fec_info(int standard, int framesize, int rate, FecInfo* out)
{
    for (const FecParamRow& r : kFecParamRows)
        if (r.standard == standard && r.framesize == framesize && r.rate == rate) {
            out->bch_k = r.bch_k; out->bch_n = r.bch_n; out->bch_t = r.bch_t;
            out->ldpc_k = r.bch_n; out->ldpc_n = r.ldpc_n;
            out->table = find_ldpc_table(r.table);
            return true;
        }
    return false;
}
int num_rates() { return (int)(sizeof(kRateNames) / sizeof(kRateNames[0])); }
const char* rate_name(int rate) { return (rate >= 0 && rate < num_rates()) ? kRateNames[rate] : nullptr; }

} // namespace dvbs2
