// bch_hip.h -- device-side BCH decoder object behind the C ABI (include/dvbs2_fec_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include <vector>

namespace dvbs2 {

// Host-side GF(2^m) tables and the BCH code parameters, built the way the reference builds them:
//   field elements by LFSR              reference lib/gf.cc:48-66
//   g(x) = product of the DISTINCT minimal polynomials of alpha^1, alpha^3, .., alpha^(2t-1)   lib/bch.cc:37-62
//   shortening s = 2^m - 1 - n, k = n - deg g                                                  lib/bch.cc:64-76
//   quadratic LUT  lut[r*r ^ r] = r for r = 0 .. 2^m - 1 (later r overwrites)                  lib/bch.cc:107-112
struct BchCode {
    int m = 0, t = 0, n = 0, k = 0, s = 0, P = 0, gdeg = 0;
    std::vector<uint16_t> antilog; // alpha^i, i in [0, P)
    std::vector<uint16_t> log;     // log[x], x in [1, P]; log[0] unused
    std::vector<uint16_t> quad;    // size P + 1
    std::vector<uint8_t> gen;      // g(x) coefficients, gen[i] = coef of x^i
    bool build(int m, uint32_t prim_poly, int t, int n, std::string* err);
};

class BchDecoderHip {
public:
    BchDecoderHip(int m, uint32_t prim_poly, int t, int n, int max_frames, int device);
    ~BchDecoderHip();
    bool ok() const { return err_.empty(); }
    // ok() reports the constructor; a failed call leaves its text in error() without disabling the handle
    const std::string& error() const { return call_err_.empty() ? err_ : call_err_; }
    const BchCode& code() const { return code_; }
    int max_frames() const { return max_frames_; }
    // DEVICE pointers. d_cw: n_frames * n/8 bytes (first bit = x^(n-1), reference lib/bch.cc:436-449);
    // d_msg: n_frames * k/8 bytes; d_corr: per frame  >= 0 corrected bits, -1 failure, -2 the reference would
    // have thrown (lib/gf.h:110 via lib/bch.cc:359-367, or lib/bch.cc:443-444).
    // d_cw == nullptr: the codewords are the hard decisions of d_llr_state (offset-binary LLR bytes, llr_stride per frame: the LDPC
    // decoder's state) -- ldpc_decoder_bb's bit packing fused into this kernel's load
    // frame_base: which range [frame_base, frame_base + n_frames) of the handle's per-frame syndrome words the call uses (the odd syndromes
    // of a batch are computed as one matrix product BEFORE the per-frame stage). Calls on disjoint ranges may be in flight together on
    // different streams (the chunks of the host-pointer chain entry); a call whose range overlaps that of an earlier call on ANOTHER
    // stream is ordered behind it with an event (two streams on one handle must not corrupt each other's syndromes).
    int decode_device(const uint8_t* d_cw, int n_frames, uint8_t* d_msg, int32_t* d_corr, hipStream_t stream,
                      const uint8_t* d_llr_state = nullptr, int llr_stride = 0, int frame_base = 0);
    // fuse bbdescrambler_bb (lib/bbdescrambler_bb_impl.cc:67-82) into the output stage: msg ^= PRBS
    int set_descramble(bool enable);

private:
    BchCode code_;
    int max_frames_, device_;
    uint16_t* d_antilog_ = nullptr;
    uint16_t* d_log_ = nullptr;
    uint16_t* d_quad_ = nullptr;
    uint16_t* d_hcol_ = nullptr;  // parity-check columns for the syndrome stage (32 B per codeword bit)
    int8_t* d_hrows_ = nullptr;   // the same matrix by rows, one byte per bit (batched syndromes on the matrix cores, bch_syndrome_kernel)
    uint32_t* d_synd_ = nullptr;  // 8 dwords per frame: the odd syndromes of the batch
    int synd_kp_ = 0, synd_rt_ = 0;
    int synd_min_frames_ = 0;     // batches of at least this many frames take the product; smaller ones the per-set-bit table
    uint8_t* d_scramble_ = nullptr;
    bool descramble_ = false;
    int n_cus_ = 0;
    size_t lds_bytes_ = 0;
    struct InFlight { hipStream_t stream = nullptr; hipEvent_t done = nullptr; int base = 0, n = 0; };
    static constexpr int kTrack = 8;
    InFlight track_[kTrack];   // the last kTrack calls: stream, syndrome range, completion event
    int track_next_ = 0;
    std::string err_;      // set by the constructor only
    std::string call_err_; // last failed call
};

// the BBFRAME energy-dispersal sequence as packed bytes (host), lib/bbdescrambler_bb_impl.cc:51-65
void bb_derandomise_sequence(uint8_t* seq, int n_bytes);

} // namespace dvbs2
