// plpayload_hip.hip -- see plpayload_hip.h. Streaming kernel, HBM bound: 8 B in + 8 B out per data symbol.
#include "plpayload_hip.h"
#include <cmath>

#include "device_guard.h"
namespace dvbs2 {

void pl_scrambling_rn(int gold_code, uint8_t* rn, int n)
{
    constexpr int P = (1 << 18) - 1;
    std::vector<uint8_t> x(P), y(P);
    for (int i = 0; i < 18; i++) { x[i] = i == 0; y[i] = 1; }
    for (int i = 0; i + 18 < P; i++) {
        x[i + 18] = x[i + 7] ^ x[i];
        y[i + 18] = y[i + 10] ^ y[i + 7] ^ y[i + 5] ^ y[i];
    }
    auto z = [&](long i) { i %= P; return (uint8_t)(x[(i + gold_code) % P] ^ y[i]); };
    for (int i = 0; i < n; i++) rn[i] = (uint8_t)(2 * z((long)i + 131072) + z(i));
}

// one thread per output (data) symbol; grid.y = frame
__global__ void pl_payload_kernel(const float2* __restrict__ in, const uint8_t* __restrict__ rn, const float* __restrict__ hphase,
                                  const float* __restrict__ pinc, const int32_t* __restrict__ cc, const float* __restrict__ pphase,
                                  float2* __restrict__ out, int n_slots, int n_pilots, int has_pilots)
{
    const int f = blockIdx.y;
    const int n_out = n_slots * 90, payload_len = n_out + n_pilots * 36;
    const bool coarse = cc[f] != 0;
    const double inc = coarse ? (double)pinc[f] : 0.0;
    for (int o = blockIdx.x * blockDim.x + threadIdx.x; o < n_out; o += gridDim.x * blockDim.x) {
        const int slot = o / 90;
        const int blk = has_pilots ? slot / 16 : 0;
        const int k = o + blk * 36; // index in the payload: pilot blocks are skipped (lib/plsync_cc_impl.cc:480-485)
        // rotator state: reset to the preceding pilot block's phase at every 16-slot segment of a coarse-corrected
        // frame (:759-763); otherwise it runs on from the PLHEADER phase (:652)
        double theta0; int steps;
        if (coarse && blk > 0) { theta0 = (double)pphase[(size_t)f * n_pilots + blk - 1]; steps = o - blk * 16 * 90; }
        else { theta0 = (double)hphase[f]; steps = o; }
        double s, c;
        sincos(-(theta0 + inc * (double)steps), &s, &c);
        const float2 x = in[(size_t)f * payload_len + k];
        float dr, di; // descrambling: multiply by conj(exp(j Rn pi/2)) in {1, -j, -1, j} (lib/pl_descrambler.cc:56-58)
        switch (rn[k]) { case 0: dr = x.x; di = x.y; break; case 1: dr = x.y; di = -x.x; break;
                         case 2: dr = -x.x; di = -x.y; break; default: dr = -x.y; di = x.x; break; }
        const float pr = (float)c, pi = (float)s;
        out[(size_t)f * n_out + o] = make_float2(dr * pr - di * pi, dr * pi + di * pr);
    }
}

PlPayloadHip::PlPayloadHip(int gold_code, int n_slots, int has_pilots, int max_frames, int device)
    : n_slots_(n_slots), has_pilots_(has_pilots ? 1 : 0), max_frames_(max_frames), device_(device)
{
    n_pilots_ = has_pilots_ ? ((n_slots_ - 1) >> 4) : 0; // lib/pl_signaling.cc:51
    if (n_slots_ < 36 || n_slots_ > 360) { err_ = "n_slots out of range (36..360)"; return; } // lib/pl_defs.h:19-20
    if (gold_code < 0 || gold_code >= (1 << 18) - 1) { err_ = "gold code out of range"; return; }
    if (max_frames_ < 1 || max_frames_ > 65535) { err_ = "max_frames must be in 1..65535 (frames are one launch dimension)"; return; }
    std::vector<uint8_t> rn(payload_len());
    pl_scrambling_rn(gold_code, rn.data(), (int)rn.size());
    DeviceGuard dev_guard(device_); // the caller's current device is restored on return
    if (!dev_guard.ok || hipMalloc(&d_rn_, rn.size()) != hipSuccess ||
        hipMemcpy(d_rn_, rn.data(), rn.size(), hipMemcpyHostToDevice) != hipSuccess) { err_ = "device setup failed"; return; }
}

PlPayloadHip::~PlPayloadHip() { DeviceGuard dev_guard(device_); (void)hipFree(d_rn_); }

int PlPayloadHip::process_device(const float* d_payload, int n_frames, const float* d_plheader_phase, const float* d_phase_inc,
                                 const int32_t* d_coarse_corrected, const float* d_pilot_phase, float* d_out, hipStream_t stream)
{
    if (!ok()) return -1;
    call_err_.clear();
    if (n_frames < 0 || n_frames > max_frames_) { call_err_ = "n_frames exceeds max_frames"; return -1; }
    if (n_frames == 0) return 0;
    DeviceGuard dev_guard(device_);
    if (!dev_guard.ok) { call_err_ = "hipSetDevice failed"; return -1; }
    hipLaunchKernelGGL(pl_payload_kernel, dim3((xfecframe_len() + 255) / 256, n_frames), dim3(256), 0, stream,
                       reinterpret_cast<const float2*>(d_payload), d_rn_, d_plheader_phase, d_phase_inc, d_coarse_corrected,
                       d_pilot_phase, reinterpret_cast<float2*>(d_out), n_slots_, n_pilots_, has_pilots_);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { call_err_ = std::string("pl payload kernel launch: ") + hipGetErrorString(e); return -1; }
    return 0;
}

} // namespace dvbs2
