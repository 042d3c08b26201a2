// bbdeheader_hip.h -- BBFRAME de-header on the device: BBHEADER check + parse, TS packet extraction across
// BBFRAME boundaries and the per-packet CRC-8 of bbdeheader_bb (reference lib/bbdeheader_bb_impl.cc:77-136 parse_bbheader,
// :138-142 check_crc8, :144-264 general_work). The block's state (d_synched, d_partial_ts_bytes, d_partial_pkt and its five
// counters) lives in device memory and is carried from call to call like the block carries it from work() to work().
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>

namespace dvbs2 {

constexpr int kBbHeaderBytes = 10; // BB_HEADER_LENGTH_BYTES
constexpr int kTsLen = 188;        // TS_PACKET_LENGTH

struct BbdhState { // device-resident block state (one per handle)
    int synched, partial;
    unsigned long long packets, errors, bbframes, dropped, gaps, overruns; // the reference's five counters + the defined deviation
    long long produced;      // bytes written by the last call
    int n_out_packets;       // packets written by the last call
    unsigned char partial_pkt[kTsLen]; // head of the TS packet that continues in the next BBFRAME
};

struct BbdhPlan { // per BBFRAME of a call, written by the scan
    int src_off;   // byte offset inside the BBFRAME of the first byte consumed by packets (after header / resync skip)
    int head;      // bytes of the first packet that come from the carried partial packet (0: none)
    int head_frame, head_off; // where those bytes are: frame index (-1: the handle's partial_pkt from the previous call), byte offset
    int n_pkts;    // packets emitted from this BBFRAME
    int out_base;  // index of its first packet in the output
};

class BbDeheaderHip {
public:
    BbDeheaderHip(int kbch_bits, int max_frames, int device);
    ~BbDeheaderHip();
    bool ok() const { return err_.empty(); }
    const std::string& error() const { return call_err_.empty() ? err_ : call_err_; }
    int kbch_bytes() const { return kbch_bytes_; }
    int max_dfl() const { return max_dfl_; }
    int max_frames() const { return max_frames_; }
    // most bytes one BBFRAME can add to the output: its DATAFIELD plus a carried partial packet, in whole packets
    int max_out_bytes_per_frame() const { return (max_dfl_ / 8 + kTsLen - 1) / kTsLen * kTsLen; }
    // DEVICE pointers: n_frames whole BBFRAMEs of kbch_bytes (the BCH decoder's descrambled messages) -> 188-byte TS packets,
    // back to back in d_out (capacity >= n_frames * max_out_bytes_per_frame()). Asynchronous on `stream`; the byte count and the
    // counters are read with state() after the stream has been synchronised.
    int process_device(const uint8_t* d_bbframes, int n_frames, uint8_t* d_out, hipStream_t stream);
    int state(BbdhState* out, hipStream_t stream); // synchronises `stream`
    int reset(hipStream_t stream);                 // the block as constructed: not synched, no partial packet, counters zero

private:
    int kbch_bytes_, max_dfl_, max_frames_, device_;
    BbdhState* d_state_ = nullptr;
    BbdhPlan* d_plan_ = nullptr;
    int* d_hdr_ = nullptr; // per frame: valid | dfl/8 << 1 | syncd/8 << 16
    std::string err_, call_err_;
};

} // namespace dvbs2
