// ldpc_schedule.h -- host-side schedule compiler for the layered LDPC kernels.
//
// The reference walks the accumulator-address table bit by bit (lib/ldpc_decoder/ldpc.hh:44-87) and
// stores, for every check node, the list of data-bit indices (lib/ldpc_decoder/layered_decoder.hh:
// 117-141), then visits checks in the order layer i = 0..q-1, lane j = 0..359 where (i, j) is the
// original check q*j + i (:135-141, :50-55). Because bit m of information group g with table address x
// is wired to check (x + m*q) mod R = q*((x div q + m) mod 360) + (x mod q), every table address is
// ONE "entry" of layer i = x mod q with shift s = x div q: check (i, j) reads data bit
// 360*g + ((j - s) mod 360). This compiler emits that (group, shift) form directly:
//
//   layer i: cnt data entries + own parity entry + previous-parity entry, each a rotated contiguous
//   360-byte window of the on-chip LLR array -- plus the sequential-order hazard information the
//   reference's strictly ordered update implies (two entries of one group in one layer), expressed
//   as a sub-block size B_i: checks may be updated 360-wide "read all, then write all" only inside
//   ascending blocks of at most B_i consecutive lanes.
#pragma once
#include <cstdint>
#include <vector>
#include "fec_tables.h"

namespace dvbs2 {

struct LdpcEntry {
    uint16_t base; // byte offset of the 360-byte window in the on-chip LLR array
    uint16_t rot;  // c = (360 - s) mod 360: lane j addresses base + (j + c) mod 360
};

struct LdpcLayer {
    uint32_t entry_off; // first entry in LdpcSchedule::entries (cnt data entries, then own parity, then previous parity)
    uint16_t cnt;       // data entries (check degree = cnt + 2; check (0,0) has no previous parity)
    uint16_t block;     // B_i: 360 when the layer has no intra-layer hazard
    uint16_t n_conflict;  // hazard layers: the first n_conflict data entries are the ones whose group occurs more
                          // than once in the layer (entry order inside a check does not affect the result)
    uint16_t sync_before; // 1: a workgroup barrier is required before this layer (it touches a group that a
                          // layer since the previous barrier also touches through a different lane mapping)
};

struct LdpcSchedule {
    const LdpcTableDesc* table = nullptr;
    int N = 0, K = 0, R = 0, q = 0;
    int cnt_max = 0;     // max data entries per layer (= LINKS_MAX_CN - 2)
    int links_total = 0; // = LINKS_TOTAL of the reference tables
    int conflict_layers = 0;
    std::vector<LdpcLayer> layers;
    std::vector<LdpcEntry> entries;
};

// Internal on-chip LLR layout: [0, K) information bits in natural order (group g = bits 360g..360g+359),
// then parity bits permuted so that pty[360*i + j] = parity[q*j + i] (layered_decoder.hh:150-152).
bool compile_ldpc_schedule(const LdpcTableDesc* t, LdpcSchedule* out);

} // namespace dvbs2
