// bbdeheader_hip.hip -- see bbdeheader_hip.h. Three launches per call:
//   1  bbdh_header_kernel   one thread per BBFRAME: CRC-8 over the ten header bytes, field checks (parse_bbheader)
//   2  bbdh_scan_kernel     the block's state machine (synched / partial count, gap and resync rules, general_work :160-247) reduced to
//                           what it decides -- where packets start, how many, where the head of a packet that straddles two BBFRAMEs
//                           lies. One workgroup: the healthy prefix of the call as two prefix sums over 256 threads (verified frame by
//                           frame), the rest (from the first gap / bad header / resync on) in order on one thread.
//   3  bbdh_packet_kernel   one workgroup per BBFRAME, one thread per TS packet: gather (at most two pieces), CRC-8, restore
//                           the sync byte, set the transport error indicator on a failed check.
// The byte work is tiny next to the decoders in front of it (kbch / 8 bytes per frame in, about as many out): no tuning beyond
// keeping everything on the device and asynchronous.
#include "bbdeheader_hip.h"
#include "device_guard.h"

namespace dvbs2 {

// remainder modulo x^8 + x^7 + x^6 + x^4 + x^2 + 1 (lib/bbdeheader_bb_impl.cc:55), one byte at a time: the register after a
// byte is the remainder of (register * x^8 + byte), i.e. table[register's contribution] folded with the incoming byte
__device__ __forceinline__ uint32_t crc8_step(uint32_t reg, uint32_t byte, const uint8_t* tab)
{
    // (reg * x^8 + byte) mod g = (reg * x^8 mod g) ^ byte  [deg(byte) < 8]; tab[r] = r * x^8 mod g
    return (uint32_t)tab[reg] ^ byte;
}
__device__ __forceinline__ void crc8_build(uint8_t* tab, int tid, int nthreads)
{
    for (int r = tid; r < 256; r += nthreads) {
        uint32_t v = (uint32_t)r << 8; // r * x^8, reduce the upper eight bits
        for (int b = 15; b >= 8; b--) if (v & (1u << b)) v ^= 0x1D5u << (b - 8);
        tab[r] = (uint8_t)v;
    }
}

__global__ void bbdh_header_kernel(const uint8_t* __restrict__ in, int n_frames, int kbch_bytes, int max_dfl, int* __restrict__ hdr)
{
    __shared__ uint8_t tab[256];
    crc8_build(tab, threadIdx.x, blockDim.x);
    __syncthreads();
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= n_frames) return;
    const uint8_t* p = in + (size_t)f * kbch_bytes;
    uint8_t h[kBbHeaderBytes];
    uint32_t reg = 0;
#pragma unroll
    for (int i = 0; i < kBbHeaderBytes; i++) { h[i] = p[i]; reg = crc8_step(reg, h[i], tab); }
    const uint32_t upl = ((uint32_t)h[2] << 8) | h[3], dfl = ((uint32_t)h[4] << 8) | h[5], syncd = ((uint32_t)h[7] << 8) | h[8];
    // parse_bbheader :80-134: CRC, dfl <= kbch - 80, dfl % 8 == 0, syncd <= dfl, upl == 188 * 8, syncd % 8 == 0
    const bool valid = reg == 0 && dfl <= (uint32_t)max_dfl && dfl % 8 == 0 && syncd <= dfl && upl == kTsLen * 8 && syncd % 8 == 0;
    hdr[f] = valid ? (int)(1u | ((dfl / 8) << 1) | ((syncd / 8) << 16)) : 0; // dfl / 8 <= 8 089 (14 bits), syncd / 8 <= 8 191
}

// The block's state machine over the frames of a call. Its state is (synched, bytes of a partial packet carried over); what a frame
// does to it depends on its header alone. In a HEALTHY stream -- every header valid, DFL >= one packet, SYNCD consistent with the
// carried bytes (general_work :194-199) -- nothing is skipped or dropped, so the carried byte count entering frame f is
//   p_in(f) = (p_in(0) + sum of DFL/8 of the frames before f) mod 188,
// frame f completes (p_in + DFL/8) / 188 packets, and the head of its first packet lies at the end of frame f - 1. That is two prefix
// sums: 256 threads compute them speculatively, verify every frame against its header (valid, DFL >= 188 bytes, SYNCD/8 == 187 - p_in
// unless p_in == 0) and write the plans of all frames BEFORE the first one that fails; from there on (a gap, a bad header, a short
// DATAFIELD, or the whole call while the block is not synchronised yet) one thread walks the frames in order exactly like round 2
// (~30 instructions per frame, started from the exact state the prefix gives). Round 2 walked every frame of every call on one thread
// (65535 frames: ~4 ms on the critical stream).
constexpr int kScanThreads = 256;
__device__ __forceinline__ int block_excl_scan(int v, int* sh, int tid, int* total)
{
    sh[tid] = v;
    __syncthreads();
    for (int d = 1; d < kScanThreads; d <<= 1) {
        const int t = tid >= d ? sh[tid - d] : 0;
        __syncthreads();
        sh[tid] += t;
        __syncthreads();
    }
    const int incl = sh[tid];
    if (total) *total = sh[kScanThreads - 1];
    __syncthreads();
    return incl - v;
}
__global__ __launch_bounds__(kScanThreads) void bbdh_scan_kernel(const int* __restrict__ hdr, int n_frames, int kbch_bytes, BbdhState* __restrict__ st, BbdhPlan* __restrict__ plan)
{
    __shared__ int sh[kScanThreads];
    __shared__ int s_first_bad, s_part, s_part_frame, s_part_off, s_out_pkts;
    const int tid = threadIdx.x;
    const int synched0 = st->synched, partial0 = st->partial;
    const int per = (n_frames + kScanThreads - 1) / kScanThreads;
    const int c0 = min(tid * per, n_frames), c1 = min(c0 + per, n_frames);
    if (tid == 0) { s_first_bad = synched0 ? n_frames : 0; s_part = partial0; s_part_frame = -1; s_part_off = 0; s_out_pkts = 0; }
    // carried bytes entering this thread's chunk (mod 188)
    int bytes = 0;
    for (int f = c0; f < c1; f++) bytes = (bytes + ((hdr[f] >> 1) & 0x7fff)) % kTsLen;
    int dummy;
    const int before = block_excl_scan(bytes, sh, tid, &dummy); // (sums of residues: at most 256 * 187)
    // verification + packets per chunk
    int p = (partial0 + before) % kTsLen, pk = 0, bad = n_frames;
    for (int f = c0; f < c1; f++) {
        const int h = hdr[f], dfl8 = (h >> 1) & 0x7fff, syncd8 = (h >> 16) & 0xffff;
        const bool ok = (h & 1) && dfl8 >= kTsLen && (p == 0 || syncd8 == kTsLen - 1 - p);
        if (!ok && bad == n_frames) bad = f;
        pk += (p + dfl8) / kTsLen;
        p = (p + dfl8) % kTsLen;
    }
    if (bad < n_frames) atomicMin(&s_first_bad, bad);
    const int pk_before = block_excl_scan(pk, sh, tid, &dummy); // (exact for every chunk that starts at or before the first failing frame)
    const int first_bad = s_first_bad;
    // plans of the healthy prefix; the thread whose chunk holds `first_bad` (or the end of the call) leaves the state there
    p = (partial0 + before) % kTsLen;
    int out_pkts = pk_before;
    for (int f = c0; f < c1 && f < first_bad; f++) {
        const int dfl8 = (hdr[f] >> 1) & 0x7fff;
        BbdhPlan pl;
        pl.src_off = kBbHeaderBytes; pl.head = p; pl.head_frame = -1; pl.head_off = 0;
        if (p > 0 && f > 0) { pl.head_frame = f - 1; pl.head_off = kBbHeaderBytes + ((hdr[f - 1] >> 1) & 0x7fff) - p; }
        pl.n_pkts = (p + dfl8) / kTsLen; pl.out_base = out_pkts;
        plan[f] = pl;
        out_pkts += pl.n_pkts;
        p = (p + dfl8) % kTsLen;
    }
    const int stop = min(first_bad, n_frames); // first frame the prefix does not cover
    if (stop > 0 && c0 < stop && stop <= c1) { // this thread wrote the plan of frame stop - 1: the state after it
        s_part = p; s_out_pkts = out_pkts;
        s_part_frame = p > 0 ? stop - 1 : -1;
        s_part_off = p > 0 ? kBbHeaderBytes + ((hdr[stop - 1] >> 1) & 0x7fff) - p : 0;
    }
    __syncthreads();
    if (tid != 0) return;
    // ---- in order from `first_bad` (general_work :160-247 reduced to what it decides per frame)
    int synched = first_bad > 0 ? 1 : synched0, partial = s_part;
    int part_frame = s_part_frame, part_off = s_part_off; // where the carried partial bytes lie (-1: st->partial_pkt)
    out_pkts = s_out_pkts;
    unsigned long long packets = (unsigned long long)out_pkts, dropped = 0, gaps = 0, overruns = 0;
    int h_next = first_bad < n_frames ? hdr[first_bad] : 0;
    for (int f = first_bad; f < n_frames; f++) {
        const int h = h_next;
        if (f + 1 < n_frames) h_next = hdr[f + 1];
        BbdhPlan pl; pl.src_off = 0; pl.head = 0; pl.head_frame = -1; pl.head_off = 0; pl.n_pkts = 0; pl.out_base = out_pkts;
        if (!(h & 1)) { synched = 0; dropped++; plan[f] = pl; continue; }          // :165-170
        int rem = (h >> 1) & 0x7fff;                                                // dfl / 8
        const int syncd8 = (h >> 16) & 0xffff;
        int off = kBbHeaderBytes;
        if (partial > 0 && syncd8 != kTsLen - 1 - partial) { synched = 0; gaps++; } // :194-199
        if (!synched) {                                                             // :203-209
            const int skip = syncd8 + 1;
            if (skip > rem) { overruns++; synched = 0; partial = 0; plan[f] = pl; continue; } // the defined deviation (bbdeheader_oracle.c)
            off += skip; rem -= skip; synched = 1; partial = 0;
        }
        pl.src_off = off;
        if (rem >= kTsLen) {                                                        // :212-238
            int n = 0;
            if (partial > 0) { pl.head = partial; pl.head_frame = part_frame; pl.head_off = part_off; rem -= kTsLen - partial; off += kTsLen - partial; partial = 0; n = 1; }
            const int whole = rem / kTsLen;
            n += whole; rem -= whole * kTsLen; off += whole * kTsLen;
            pl.n_pkts = n; out_pkts += n; packets += (unsigned long long)n;
        }
        if (rem > 0) { partial = rem; part_frame = f; part_off = off; }             // :241-245 (a partial that could not be completed is replaced)
        plan[f] = pl;
    }
    st->synched = synched; st->partial = partial;
    st->packets += packets; st->bbframes += (unsigned long long)n_frames; st->dropped += dropped; st->gaps += gaps; st->overruns += overruns;
    st->n_out_packets = out_pkts; st->produced = (long long)out_pkts * kTsLen;
    // the carried bytes themselves are saved by the packet kernel (last workgroup): frame index + offset travel in the plan slot
    // one past the last frame
    BbdhPlan tail; tail.src_off = part_off; tail.head = partial; tail.head_frame = part_frame; tail.head_off = 0; tail.n_pkts = 0; tail.out_base = out_pkts;
    plan[n_frames] = tail;
}

// one workgroup per BBFRAME, one thread per TS packet (at most 39 per frame: 7 264 DATAFIELD bytes + 187 carried)
__global__ void bbdh_packet_kernel(const uint8_t* __restrict__ in, int n_frames, int kbch_bytes, BbdhState* __restrict__ st,
                                   const BbdhPlan* __restrict__ plan, uint8_t* __restrict__ out)
{
    __shared__ uint8_t tab[256];
    __shared__ uint8_t carried[kTsLen];
    __shared__ int bad_count;
    const int f = blockIdx.x, tid = threadIdx.x;
    const BbdhPlan pl = plan[f];
    crc8_build(tab, tid, blockDim.x);
    if (tid == 0) bad_count = 0;
    if (pl.head > 0) {
        const uint8_t* src = pl.head_frame < 0 ? st->partial_pkt : in + (size_t)pl.head_frame * kbch_bytes + pl.head_off;
        for (int i = tid; i < pl.head; i += blockDim.x) carried[i] = src[i];
    }
    __syncthreads();
    if (tid < pl.n_pkts) {
        const uint8_t* frame = in + (size_t)f * kbch_bytes;
        uint8_t* o = out + ((size_t)pl.out_base + tid) * kTsLen;
        // packet tid: bytes [0, 188); with a carried head the first packet takes `head` bytes from it and the rest from the frame
        int first_from_frame = 0;             // bytes of packet 0 that come from `carried`
        int off;                              // frame offset of this packet's first frame byte
        if (pl.head > 0) {
            if (tid == 0) { first_from_frame = pl.head; off = pl.src_off; }
            else off = pl.src_off + (kTsLen - pl.head) + (tid - 1) * kTsLen;
        } else off = pl.src_off + tid * kTsLen;
        uint32_t reg = 0;
        uint32_t prev = 0x47; // byte to write at position i: out[0] = sync byte, out[i] = packet[i - 1]
        for (int i = 0; i < kTsLen; i++) {
            const uint32_t b = i < first_from_frame ? carried[i] : frame[off + i - first_from_frame];
            reg = crc8_step(reg, b, tab);
            o[i] = (uint8_t)prev;
            prev = b;
        }
        // (the packet's last byte is the CRC that sits in the next packet's sync position: checked, not copied)
        if (reg != 0) { o[1] |= 0x80; atomicAdd(&bad_count, 1); } // TRANSPORT_ERROR_INDICATOR :228-232
    }
    __syncthreads();
    if (tid == 0 && bad_count) atomicAdd(&st->errors, (unsigned long long)bad_count);
}

// the partial packet for the next call (a launch of its own, after every reader of the old one has finished)
__global__ void bbdh_save_partial_kernel(const uint8_t* __restrict__ in, int n_frames, int kbch_bytes, BbdhState* __restrict__ st, const BbdhPlan* __restrict__ plan)
{
    const BbdhPlan t = plan[n_frames];
    if (t.head_frame < 0) return; // unchanged (or empty)
    const int tid = threadIdx.x;
    if (tid < t.head) st->partial_pkt[tid] = in[(size_t)t.head_frame * kbch_bytes + t.src_off + tid];
}

BbDeheaderHip::BbDeheaderHip(int kbch_bits, int max_frames, int device)
    : kbch_bytes_(kbch_bits / 8), max_dfl_(kbch_bits - 80), max_frames_(max_frames), device_(device)
{
    if (kbch_bits < 88 || kbch_bits % 8 != 0 || kbch_bits - 80 > 0xffff) { err_ = "unsupported BCH message length"; return; }
    if (max_frames_ < 1 || max_frames_ > 65535) { err_ = "max_frames must be in 1..65535"; return; }
    if (max_out_bytes_per_frame() / kTsLen > 64) { err_ = "more than 64 packets per BBFRAME"; return; }
    DeviceGuard guard(device_);
    if (!guard.ok) { err_ = "hipSetDevice failed"; return; }
    hipError_t e = hipMalloc(&d_state_, sizeof(BbdhState));
    if (e == hipSuccess) e = hipMemset(d_state_, 0, sizeof(BbdhState));
    if (e == hipSuccess) e = hipMalloc(&d_plan_, (size_t)(max_frames_ + 1) * sizeof(BbdhPlan));
    if (e == hipSuccess) e = hipMalloc(&d_hdr_, (size_t)max_frames_ * 4);
    if (e != hipSuccess) err_ = std::string("bbdeheader buffers: ") + hipGetErrorString(e);
}

BbDeheaderHip::~BbDeheaderHip()
{
    DeviceGuard guard(device_);
    (void)hipFree(d_state_); (void)hipFree(d_plan_); (void)hipFree(d_hdr_);
}

int BbDeheaderHip::process_device(const uint8_t* d_bbframes, int n_frames, uint8_t* d_out, hipStream_t stream)
{
    if (!ok()) return -1;
    call_err_.clear();
    if (n_frames < 0 || n_frames > max_frames_) { call_err_ = "n_frames exceeds max_frames"; return -1; }
    DeviceGuard guard(device_);
    if (!guard.ok) { call_err_ = "hipSetDevice failed"; return -1; }
    hipLaunchKernelGGL(bbdh_header_kernel, dim3((n_frames + 255) / 256 + (n_frames == 0)), dim3(256), 0, stream, d_bbframes, n_frames, kbch_bytes_, max_dfl_, d_hdr_);
    hipLaunchKernelGGL(bbdh_scan_kernel, dim3(1), dim3(kScanThreads), 0, stream, d_hdr_, n_frames, kbch_bytes_, d_state_, d_plan_);
    if (n_frames > 0) {
        hipLaunchKernelGGL(bbdh_packet_kernel, dim3(n_frames), dim3(64), 0, stream, d_bbframes, n_frames, kbch_bytes_, d_state_, d_plan_, d_out);
        hipLaunchKernelGGL(bbdh_save_partial_kernel, dim3(1), dim3(192), 0, stream, d_bbframes, n_frames, kbch_bytes_, d_state_, d_plan_);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { call_err_ = std::string("bbdeheader launch: ") + hipGetErrorString(e); return -1; }
    return 0;
}

int BbDeheaderHip::state(BbdhState* out, hipStream_t stream)
{
    if (!ok()) return -1;
    call_err_.clear();
    DeviceGuard guard(device_);
    if (!guard.ok) { call_err_ = "hipSetDevice failed"; return -1; }
    hipError_t e = hipMemcpyAsync(out, d_state_, sizeof(BbdhState), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    if (e != hipSuccess) { call_err_ = std::string("bbdeheader state: ") + hipGetErrorString(e); return -1; }
    return 0;
}

int BbDeheaderHip::reset(hipStream_t stream)
{
    if (!ok()) return -1;
    call_err_.clear();
    DeviceGuard guard(device_);
    if (!guard.ok) { call_err_ = "hipSetDevice failed"; return -1; }
    hipError_t e = hipMemsetAsync(d_state_, 0, sizeof(BbdhState), stream);
    if (e != hipSuccess) { call_err_ = std::string("bbdeheader reset: ") + hipGetErrorString(e); return -1; }
    return 0;
}

} // namespace dvbs2
