// demap_hip.h -- soft constellation demapper (QPSK, 8PSK) behind the C ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include "demap_math.hpp"

namespace dvbs2 {

class DemapperHip {
public:
    // framesize / rate / constellation: reference enums (dvb_config.h). Mirrors the constructor of
    // xfecframe_demapper_cb_impl (lib/xfecframe_demapper_cb_impl.cc:27-91): frame length by framesize,
    // QPSK or 8PSK only ("Unsupported constellation" otherwise, :70-72), 8PSK column order by rate (:50-69).
    DemapperHip(int framesize, int rate, int constellation, int max_frames, int device);
    bool ok() const { return err_.empty(); }
    // ok() reports the constructor; a failed call leaves its text in error() without disabling the handle
    const std::string& error() const { return call_err_.empty() ? err_ : call_err_; }
    int n_llr() const { return n_llr_; }
    int n_mod() const { return n_mod_; }
    int n_syms() const { return n_llr_ / n_mod_; }
    int column_order() const { return order_; }
    int max_frames() const { return max_frames_; }
    // DEVICE pointers. syms: n_frames * n_syms interleaved (re, im) floats; n0: n0_count (1 or n_frames) noise
    // energies N0 (the block's d_N0); llr_out: n_frames * n_llr int8, de-interleaved for 8PSK.
    int soft_device(const float* d_syms, int n_frames, const float* d_n0, int n0_count, int8_t* d_llr, hipStream_t stream);
    // pre-decoder linear SNR per frame (lib/xfecframe_demapper_cb_impl.cc:128-149): float reduction,
    // order differs from the reference's sequential / VOLK accumulation -> tolerance only.
    // d_ref_llr != nullptr: post-decoder refinement against the decoded LLRs (:246-317)
    int snr_device(const float* d_syms, const int8_t* d_ref_llr, int n_frames, float* d_snr, hipStream_t stream);
    // what an LDPC sweep kernel needs to do this demapper's work while it loads its frames (same arithmetic: demap_math.hpp)
    DemapFused fused(const float* d_syms, const float* d_n0, int n0_count) const;

private:
    int n_llr_ = 0, n_mod_ = 0, order_ = 0, constellation_ = 0, max_frames_ = 0, device_ = 0;
    std::string err_;      // set by the constructor only
    std::string call_err_; // last failed call
};

} // namespace dvbs2
