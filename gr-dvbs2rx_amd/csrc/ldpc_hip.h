// ldpc_hip.h -- device-side LDPC decoder object behind the C ABI (include/dvbs2_fec_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include "ldpc_schedule.h"
#include "demap_math.hpp"

namespace dvbs2 {

// per-CU wave-pattern counters of the one-frame sweep kernels: one device array per device key (ldpc_hip.hip)
int* cu_slot_table(int device_key, std::string* err);
constexpr int kCuSlotWords = 16 * 8 * 2 * 16;

class LdpcDecoderHip {
public:
    // group_size G: frames [G*g, G*g+G) share one iteration count, exactly like one SIMD batch of the
    // reference (lib/ldpc_decoder_bb_impl.cc:406-410, lib/ldpc_decoder/layered_decoder.hh:153). G = 1
    // stops every frame on its own.
    LdpcDecoderHip(const LdpcTableDesc* table, int out_bits_message, int group_size, int max_frames, int device);
    ~LdpcDecoderHip();
    bool ok() const { return err_.empty(); }
    // ok() reports the constructor; a failed call leaves its text in error() without disabling the handle
    const std::string& error() const { return call_err_.empty() ? err_ : call_err_; }

    int N() const { return sched_.N; }
    int K() const { return sched_.K; }
    int q() const { return sched_.q; }
    int out_bits_message() const { return out_bits_message_; }
    // name of the sweep kernel this handle launches (as a profiler shows it, without the argument list)
    const char* kernel_name() const { return kname_.c_str(); }
    int group_size() const { return G_; }
    int max_frames() const { return max_frames_; }
    const LdpcSchedule& schedule() const { return sched_; }

    // All pointers are DEVICE pointers. d_llr_in: n_frames*N int8 (frame-major, positive = bit 0).
    // d_bits_out: n_frames * (out_mode ? out_bits_message/8 : N/8) bytes, MSB first.
    // d_llr_out (nullable): n_frames*N decoded LLRs (what the reference publishes as llr_pdu).
    // d_ret (nullable): one int32 per group = the reference decode() return value (trials left, or -1).
    //
    // enqueue() puts the whole decode on `stream` WITHOUT any host synchronisation: the first pass, kResolveRounds
    // rounds of (group targets, resume launch -- workgroups with nothing to do leave at once) that resolve the
    // batch-coupled stopping rule on the device, the output stage, and an asynchronous read-back of the "groups still
    // unresolved" counter. finish() waits for the stream and, only if that counter is not zero (a group needed more
    // rounds than were enqueued: a frame that had converged earlier fails the test again at the group's count -- rare),
    // runs the remaining rounds and the output stage again. Results are final after finish() returned 0.
    // `slot` (< kSlots) selects the per-call flag storage and `frame_base` the range [frame_base, frame_base + n_frames)
    // of the handle's state/message buffers, so that several chunks can be in flight on different streams.
    // decode_device() = enqueue() + finish().
    static constexpr int kResolveRounds = 2;
    static constexpr int kSlots = 4;
    // dm (optional): take XFECFRAME symbols instead of LLRs and demap while the frames are loaded (d_llr_in is ignored then);
    // only the classic sweep kernels do that (fused_demap_supported()). d_bits_out may be null when neither packed bits nor
    // decoded LLRs are wanted (the chain's BCH stage reads the decoder state directly: state(), state_stride()).
    int enqueue(const int8_t* d_llr_in, int n_frames, int max_trials, int out_mode, uint8_t* d_bits_out, int8_t* d_llr_out,
                int32_t* d_ret, hipStream_t stream, int slot = 0, int frame_base = 0, const DemapFused* dm = nullptr);
    bool fused_demap_supported() const { return !pr_; }
    const uint8_t* state() const { return d_state_; } // per frame N offset-binary LLR bytes, information part in natural order

    int finish(int slot = 0); // 0 = done, 1 = done and the outputs were rewritten by extra rounds, -1 = error
    // error paths of the callers: wait for whatever is in flight on every slot and free the slots (results are discarded; the
    // error text of the failed call is kept)
    void abort_all();
    int decode_device(const int8_t* d_llr_in, int n_frames, int max_trials, int out_mode,
                      uint8_t* d_bits_out, int8_t* d_llr_out, int32_t* d_ret, hipStream_t stream);

    // average duration (ms) and launch count of the main update kernel since the last reset (HIP events
    // on the launch stream; only recorded when profiling is enabled).
    void set_profiling(bool on) { profiling_ = on; }
    void reset_profile() { prof_ms_ = 0; prof_launches_ = 0; }
    double profile_ms() const { return prof_ms_; }
    int profile_launches() const { return prof_launches_; }
    // host-driven resolution rounds since the handle was created (finish(): a waiting frame of the group-synchronous stop gave up
    // and the group was completed by resume launches): zero in normal operation, asserted by the tests and printed by bench.py
    int fallback_rounds() const { return fallback_rounds_; }

private:
    LdpcSchedule sched_;
    int out_bits_message_, G_, max_frames_, device_;
    int words_per_check_ = 0; // message dwords per check (4 int8 messages per dword)
    int dmax_ = 0;            // kernel variant: handles check degrees dmax-7 .. dmax (8, 12, ..., 32)
    uint32_t* d_recs_ = nullptr;  // per-layer records (ldpc_hip.hip); kRecHeaderWords of header in front of them (d_recs_alloc_)
    uint32_t* d_recs_alloc_ = nullptr;
    uint32_t* d_wrecs_ = nullptr; // per-(layer, wave) sweep records of the classic kernel
    size_t lds_bytes_ = 0;
    std::string kname_;
    bool v2_ = false;             // the build with the packed nodes (check_node_v2 / check_node_chain_v2)
    bool chain_plain_ = false;    // plain sweep kernel with the packed register chain for single-pair hazard layers
    bool soft_bar_ = false;       // per-frame software barriers (high-degree tables without hazard layers)
    int fallback_rounds_ = 0;
    bool pr_w1_ = false;          // parity-in-records kernel with one-dword records (check degree <= 4)
    bool pr_v2_ = false;          // parity-in-records kernel with packed nodes in the regular middle layers (two-dword records, per-wave sweep records)
    bool hz2_ = false;            // the build with the heavy-hazard paths (ldpc_kernel.hpp, HZ2)
    bool solo_ = false;           // one frame per workgroup, complementary wave roles per CU (ldpc_kernel.hpp)
    int* d_cu_slots_ = nullptr;   // per-CU pattern counters of the solo kernels
    bool dense_ = false;          // 80-VGPR build of the classic kernel selected (two workgroups per CU)
    bool pr_ = false;             // parity-in-records kernel variant selected (ldpc_kernel_pr.hpp)
    unsigned long long* d_tdbg_ = nullptr; // DVBS2_TIMING=1: per-wave cycle-counter breakdown (diagnostics)
    uint8_t* d_state_ = nullptr;  // max_frames * N, internal layout, offset-binary LLRs
    uint32_t* d_msgs_ = nullptr;  // max_frames * q * words_per_check * 384
    int* d_iters_ = nullptr;      // per frame: updates done
    int* d_good_ = nullptr;       // per frame: syndrome satisfied at the current state
    int* d_target_ = nullptr;     // per frame: updates to reach in a resume pass
    int* d_gsync_ = nullptr;      // one status word per frame: group-synchronous stop inside the first pass (ldpc_kernel.hpp, group_decide)
    bool gsync_on_ = false;
    bool pr_shared_sv_ = false;   // parity-in-records kernel: one sign-vector area per workgroup (two do not fit twice into a CU's LDS)
    int* d_flag_ = nullptr;       // [slot] = number of unresolved groups
    int* h_flag_ = nullptr;       // pinned, [slot]
    struct Pending { bool active = false; int n_frames = 0, max_trials = 0, out_mode = 0, frame_base = 0;
                     uint8_t* bits = nullptr; int8_t* llr_out = nullptr; int32_t* ret = nullptr; hipStream_t stream = nullptr; };
    Pending pend_[kSlots];
    int resolve_rounds_ = kResolveRounds;
    void launch_sweep(const int8_t* in, bool resume, int stop_on_good, int n_frames, int max_trials, int frame_base, hipStream_t stream, const DemapFused* dm = nullptr);
    void launch_targets(int n_frames, int max_trials, int frame_base, int32_t* d_ret, int slot, hipStream_t stream);
    void launch_finalize(const Pending& p);
    bool profiling_ = false;
    double prof_ms_ = 0;
    int prof_launches_ = 0;
    hipEvent_t ev0_ = nullptr, ev1_ = nullptr;
    std::string err_;      // set by the constructor only
    std::string call_err_; // last failed call
};

} // namespace dvbs2
