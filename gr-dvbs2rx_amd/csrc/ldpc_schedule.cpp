// ldpc_schedule.cpp -- see ldpc_schedule.h.
#include "ldpc_schedule.h"
#include <algorithm>
#include <set>

namespace dvbs2 {

bool compile_ldpc_schedule(const LdpcTableDesc* t, LdpcSchedule* out)
{
    if (!t) return false;
    LdpcSchedule& s = *out;
    s = LdpcSchedule();
    s.table = t;
    s.N = t->N; s.K = t->K; s.R = s.N - s.K; s.q = s.R / 360;
    if (s.K != t->nrows * 360 || s.R != s.q * 360) return false;

    struct GS { int g, sh; };
    std::vector<std::vector<GS>> per_layer(s.q);
    const uint16_t* p = ldpc_table_words(t);
    long links = 0;
    for (int g = 0; g < t->nrows; g++) {
        int deg = *p++;
        for (int n = 0; n < deg; n++) {
            int x = p[n];
            if (x >= s.R) return false;
            per_layer[x % s.q].push_back({ g, x / s.q });
        }
        p += deg;
        links += 360L * deg;
    }
    s.links_total = (int)(links + 2L * s.R - 1);

    // Barrier elision: parity rows are always touched by the same thread (rot 0) except row q-1 in layer 0,
    // so only the DATA groups (and that one row) can carry a cross-thread hazard between layers.
    std::set<int> epoch;
    for (int i = 0; i < s.q; i++) {
        auto& v = per_layer[i];
        LdpcLayer L;
        L.entry_off = (uint32_t)s.entries.size();
        L.cnt = (uint16_t)v.size();
        s.cnt_max = std::max<int>(s.cnt_max, L.cnt);
        int block = 360;
        for (size_t a = 0; a < v.size(); a++)
            for (size_t b = a + 1; b < v.size(); b++)
                if (v[a].g == v[b].g) {
                    int d = ((v[b].sh - v[a].sh) % 360 + 360) % 360;
                    if (d == 0) return false; // a check would touch the same bit twice
                    block = std::min(block, std::min(d, 360 - d));
                }
        L.block = (uint16_t)block;
        if (block < 360) s.conflict_layers++;
        std::set<int> mine;
        for (const GS& e : v) mine.insert(e.g);
        if (i == 0) mine.insert(t->nrows + s.q - 1);       // previous-parity row of layer 0 (rot 359)
        if (i == s.q - 1) mine.insert(t->nrows + s.q - 1); // the same row, as own parity of the last layer
        bool hit = (i == 0) || block < 360;
        for (int g : mine) if (epoch.count(g)) hit = true;
        L.sync_before = hit ? 1 : 0;
        if (hit) epoch.clear();
        epoch.insert(mine.begin(), mine.end());

        // hazard entries first
        std::vector<GS> ordered;
        int n_conf = 0;
        for (int pass = 0; pass < 2; pass++)
            for (size_t a = 0; a < v.size(); a++) {
                int same = 0;
                for (size_t b = 0; b < v.size(); b++) same += v[b].g == v[a].g;
                if ((same > 1) == (pass == 0)) { ordered.push_back(v[a]); n_conf += pass == 0; }
            }
        L.n_conflict = (uint16_t)n_conf;
        for (const GS& e : ordered)
            s.entries.push_back({ (uint16_t)(360 * e.g), (uint16_t)((360 - e.sh) % 360) });
        // own parity pty[360*i + j]
        s.entries.push_back({ (uint16_t)(s.K + 360 * i), 0 });
        // previous parity: pty[360*(i-1) + j], or pty[360*(q-1) + j - 1] for layer 0 (lane 0 has none)
        if (i) s.entries.push_back({ (uint16_t)(s.K + 360 * (i - 1)), 0 });
        else s.entries.push_back({ (uint16_t)(s.K + 360 * (s.q - 1)), 359 });
        s.layers.push_back(L);
    }
    return true;
}

} // namespace dvbs2
