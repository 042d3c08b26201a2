// plpayload_hip.h -- PLFRAME payload step on the device: PL descrambling, pilot removal and the
// per-segment phase de-rotation that plsync_cc_impl::handle_payload() applies before the symbols reach
// xfecframe_demapper_cb (reference lib/plsync_cc_impl.cc:644-653, :727-795; lib/pl_descrambler.cc).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include <vector>

namespace dvbs2 {

// Rn(i) in 0..3, i < n, from the definition in ETSI EN 302 307-1 clause 5.5.4: x(i+18) = x(i+7) + x(i), x(0) = 1;
// y(i+18) = y(i+10) + y(i+7) + y(i+5) + y(i), y(0..17) = 1; z_n(i) = x((i + n) mod (2^18 - 1)) + y(i);
// Rn(i) = 2 z_n((i + 131072) mod (2^18 - 1)) + z_n(i). (The reference reaches the same numbers with register masks.)
void pl_scrambling_rn(int gold_code, uint8_t* rn, int n);

class PlPayloadHip {
public:
    PlPayloadHip(int gold_code, int n_slots, int has_pilots, int max_frames, int device);
    ~PlPayloadHip();
    bool ok() const { return err_.empty(); }
    // ok() reports the constructor; a failed call leaves its text in error() without disabling the handle
    const std::string& error() const { return call_err_.empty() ? err_ : call_err_; }
    int n_slots() const { return n_slots_; }
    int n_pilots() const { return n_pilots_; }
    int payload_len() const { return n_slots_ * 90 + n_pilots_ * 36; }
    int xfecframe_len() const { return n_slots_ * 90; }
    int max_frames() const { return max_frames_; }
    // DEVICE pointers. d_payload: n_frames * payload_len complex (re, im); per frame: PLHEADER phase, phase increment
    // per symbol (2 pi fine_foffset, 0 when the frame is not coarse-corrected), coarse-corrected flag, n_pilots pilot
    // phases (ignored without pilots / when not coarse-corrected); d_out: n_frames * xfecframe_len complex.
    int process_device(const float* d_payload, int n_frames, const float* d_plheader_phase, const float* d_phase_inc,
                       const int32_t* d_coarse_corrected, const float* d_pilot_phase, float* d_out, hipStream_t stream);

private:
    int n_slots_, n_pilots_, has_pilots_, max_frames_, device_;
    uint8_t* d_rn_ = nullptr;
    std::string err_;      // set by the constructor only
    std::string call_err_; // last failed call
};

} // namespace dvbs2
