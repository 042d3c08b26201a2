// dump_hazards.cc -- host-only: hazard structure of every layer of a table (which groups occur more than once, row distances).
// g++ -O2 -std=c++17 -Igr-dvbs2rx_amd/csrc tools/dump_hazards.cc gr-dvbs2rx_amd/csrc/fec_tables.cpp gr-dvbs2rx_amd/csrc/ldpc_schedule.cpp -o tools/bin/dump_hazards
#include "ldpc_schedule.h"
#include <cstdio>
#include <cstring>
#include <map>
#include <vector>
#include <algorithm>
using namespace dvbs2;
int main(int argc, char** argv)
{
    for (int a = 1; a < argc; a++) {
        const LdpcTableDesc* t = find_ldpc_table(argv[a]);
        if (!t) { fprintf(stderr, "no table %s\n", argv[a]); continue; }
        LdpcSchedule s;
        compile_ldpc_schedule(t, &s);
        printf("%s N %d K %d q %d hazard layers %d\n", t->name, s.N, s.K, s.q, s.conflict_layers);
        for (int i = 0; i < s.q; i++) {
            const LdpcLayer& L = s.layers[i];
            if (L.block >= 360) continue;
            std::map<int, std::vector<int>> by_group;
            for (int k = 0; k < L.n_conflict; k++) { const LdpcEntry& e = s.entries[L.entry_off + k]; by_group[e.base / 360].push_back(e.rot); }
            printf("  layer %2d deg %2d nconf %d block %3d:", i, L.cnt + 2, L.n_conflict, L.block);
            for (auto& g : by_group) {
                printf(" g%d{", g.first);
                for (size_t x = 0; x < g.second.size(); x++) printf("%s%d", x ? "," : "", g.second[x]);
                printf("}d=");
                bool first = true;
                for (size_t x = 0; x < g.second.size(); x++) for (size_t y = x + 1; y < g.second.size(); y++) {
                    int d = std::abs(g.second[x] - g.second[y]); d = std::min(d, 360 - d);
                    printf("%s%d", first ? "" : ",", d); first = false;
                }
            }
            printf("\n");
        }
    }
    return 0;
}
