// tools/extract_ldpc_tables.cc -- RUN IN THE BUILD CONTAINER ONLY (needs /root/reference).
//
// Dumps the DVB-S2 / S2X / T2 LDPC parity-check address tables (ETSI EN 302 307-1 Annex B/C,
// EN 302 307-2 Annex B/C, EN 302 755 Annex A/B -- standard-defined constants) that the
// reference keeps in lib/dvb_s2_tables.hh, lib/dvb_s2x_tables.hh and lib/dvb_t2_tables.hh
// into this repo's own compact row format:
//
//   per table:  name N K nrows  then nrows lines "deg a0 a1 ... a(deg-1)"
//
// (one row per 360-bit group of information bits). The output is DATA; the repo's own
// schedule compiler (csrc/ldpc_schedule.cpp) turns it into the layered (group, shift)
// schedule. The generator is committed so that the provenance of the numbers is auditable.
#include <cstdio>
#include <cstdlib>
#include "dvb_s2_tables.hh"
#include "dvb_s2x_tables.hh"
#include "dvb_t2_tables.hh"

template <typename T>
static void dump(const char* name)
{
    int nrows = 0;
    for (int g = 0; T::LEN[g]; ++g) nrows += T::LEN[g];
    std::printf("T %s %d %d %d\n", name, T::N, T::K, nrows);
    const int* p = T::POS;
    long links = 0;
    for (int g = 0; T::LEN[g]; ++g)
        for (int r = 0; r < T::LEN[g]; ++r) {
            std::printf("%d", T::DEG[g]);
            for (int c = 0; c < T::DEG[g]; ++c) std::printf(" %d", *p++);
            std::printf("\n");
            links += T::DEG[g];
        }
    // sanity: LINKS_TOTAL = 360*links (data) + 2*R - 1 (parity zig-zag)
    long expect = 360L * links + 2L * (T::N - T::K) - 1;
    if (expect != T::LINKS_TOTAL || nrows * 360 != T::K) {
        std::fprintf(stderr, "table %s inconsistent (%ld vs %d)\n", name, expect, T::LINKS_TOTAL);
        std::exit(1);
    }
}
#define D(x) dump<DVB_##x>(#x)
int main()
{
    D(S2_TABLE_B1); D(S2_TABLE_B2); D(S2_TABLE_B3); D(S2_TABLE_B4); D(S2_TABLE_B5); D(S2_TABLE_B6);
    D(S2_TABLE_B7); D(S2_TABLE_B8); D(S2_TABLE_B9); D(S2_TABLE_B10); D(S2_TABLE_B11);
    D(S2_TABLE_C1); D(S2_TABLE_C2); D(S2_TABLE_C3); D(S2_TABLE_C4); D(S2_TABLE_C5); D(S2_TABLE_C6);
    D(S2_TABLE_C7); D(S2_TABLE_C8); D(S2_TABLE_C9); D(S2_TABLE_C10);
    D(S2X_TABLE_B1); D(S2X_TABLE_B2); D(S2X_TABLE_B3); D(S2X_TABLE_B4); D(S2X_TABLE_B5); D(S2X_TABLE_B6);
    D(S2X_TABLE_B7); D(S2X_TABLE_B8); D(S2X_TABLE_B9); D(S2X_TABLE_B10); D(S2X_TABLE_B11); D(S2X_TABLE_B12);
    D(S2X_TABLE_B13); D(S2X_TABLE_B14); D(S2X_TABLE_B15); D(S2X_TABLE_B16); D(S2X_TABLE_B17); D(S2X_TABLE_B18);
    D(S2X_TABLE_B19); D(S2X_TABLE_B20); D(S2X_TABLE_B21); D(S2X_TABLE_B22); D(S2X_TABLE_B23); D(S2X_TABLE_B24);
    D(S2X_TABLE_C1); D(S2X_TABLE_C2); D(S2X_TABLE_C3); D(S2X_TABLE_C4); D(S2X_TABLE_C5); D(S2X_TABLE_C6);
    D(S2X_TABLE_C7); D(S2X_TABLE_C8); D(S2X_TABLE_C9); D(S2X_TABLE_C10);
    D(T2_TABLE_A3); D(T2_TABLE_B3);
    return 0;
}
