// c_api.hip -- extern "C" boundary (include/dvbs2_fec_hip.h). No exceptions leave this file.
#include "../../include/dvbs2_fec_hip.h"
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
#include <new>
#include <string>
#include <utility>
#include <vector>
#include "fec_tables.h"
#include "ldpc_hip.h"
#include "ldpc_schedule.h"
#include "bch_hip.h"
#include "demap_hip.h"
#include "plpayload_hip.h"
#include "bbdeheader_hip.h"
#include "device_guard.h"
#include "demap_math.hpp"
#include <algorithm>

using namespace dvbs2;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define HCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(DVBS2_EDEVICE, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
#define API_TRY try {
#define API_CATCH } catch (const std::exception& e) { return fail(DVBS2_EDEVICE, e.what()); } catch (...) { return fail(DVBS2_EDEVICE, "unknown exception"); }

struct dvbs2_ldpc {
    LdpcDecoderHip* dec = nullptr;
    // host staging (dvbs2_ldpc_decode): device copies of the caller's buffers and two streams, so that the transfer of
    // chunk c + 1 runs under the decode of chunk c
    int8_t* d_in = nullptr; uint8_t* d_bits = nullptr; int8_t* d_llr = nullptr; int32_t* d_ret = nullptr;
    // pinned landing buffers for the outputs: a device-to-host copy into pageable memory blocks the calling thread until
    // the chunk's kernels are done, which would serialise the chunks
    uint8_t* p_bits = nullptr; int8_t* p_llr = nullptr; int32_t* p_ret = nullptr;
    hipStream_t stream[LdpcDecoderHip::kSlots] = {};
    // inputs go through ONE copy stream, chunk after chunk (each copy at the full link rate, chunk 0 first), and the chunk's compute
    // stream waits for its event: with the copies on the four compute streams a page-locked caller's chunks 0..3 shared the link and
    // the first kernel started after FOUR chunks had arrived instead of one (round 3: page-locked callers 8 % slower than pageable ones)
    hipStream_t copy_stream = nullptr;
    hipEvent_t in_ready[LdpcDecoderHip::kSlots] = {};
    int device = 0;
    // experiment / test knobs of the host entry, read ONCE when the handle is created (no getenv per decode call)
    std::string host_plan;      // DVBS2_HOST_PLAN: comma list of chunk sizes, the last one repeats
    int host_chunk = 0;         // DVBS2_HOST_CHUNK: one chunk size for the whole call (0: the measured plan)
    int host_copy_stream = -1;  // DVBS2_HOST_COPY_STREAM: 0 / 1 force the copies off / onto the copy stream (-1: by kind of input buffer)
};

// Is [p, p + bytes) ONE page-locked host range the copy engine can address (hipHostMalloc'ed, or registered with dvbs2_host_register /
// hipHostRegister)? Decided from the runtime's own record of the allocation that holds p -- its start and size
// (HIP_POINTER_ATTRIBUTE_RANGE_START_ADDR / RANGE_SIZE) must cover the last byte -- so two registrations with a pageable hole between
// them are not mistaken for one. When the runtime cannot answer, the range counts as pageable: the call then stages through the
// handle's own pinned buffers, which is always correct.
static bool host_range_page_locked(const void* p, size_t bytes)
{
    if (!p || !bytes) return false;
    hipPointerAttribute_t a{};
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; } // an ordinary pageable pointer: not an error
    if (a.type != hipMemoryTypeHost) return false;
    void* start = nullptr; size_t size = 0;
    if (hipPointerGetAttribute(&start, HIP_POINTER_ATTRIBUTE_RANGE_START_ADDR, const_cast<void*>(p)) != hipSuccess ||
        hipPointerGetAttribute(&size, HIP_POINTER_ATTRIBUTE_RANGE_SIZE, const_cast<void*>(p)) != hipSuccess || !start || !size) {
        (void)hipGetLastError();
        return false;
    }
    const char* lo = (const char*)start; const char* q = (const char*)p;
    return q >= lo && (size_t)(q - lo) + bytes <= size;
}

extern "C" {

const char* dvbs2_last_error(void) { return g_err.c_str(); }

int dvbs2_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int dvbs2_host_register(void* p, size_t bytes)
{
    if (!p || !bytes) return fail(DVBS2_EINVAL, "bad argument");
    HCHK(hipHostRegister(p, bytes, hipHostRegisterDefault));
    return DVBS2_OK;
}

int dvbs2_host_is_page_locked(const void* p, size_t bytes) { return host_range_page_locked(p, bytes) ? 1 : 0; }

int dvbs2_host_alloc(void** p, size_t bytes)
{
    if (!p || !bytes) return fail(DVBS2_EINVAL, "bad argument");
    *p = nullptr;
    HCHK(hipHostMalloc(p, bytes, hipHostMallocDefault));
    return DVBS2_OK;
}

int dvbs2_host_free(void* p)
{
    if (!p) return DVBS2_OK;
    HCHK(hipHostFree(p));
    return DVBS2_OK;
}

int dvbs2_host_unregister(void* p)
{
    if (!p) return fail(DVBS2_EINVAL, "bad argument");
    HCHK(hipHostUnregister(p));
    return DVBS2_OK;
}

int dvbs2_get_fec_info(int standard, int framesize, int rate, dvbs2_fec_info_t* out)
{
    if (!out) return fail(DVBS2_EINVAL, "null out");
    FecInfo fi;
    if (!get_fec_info(standard, framesize, rate, &fi)) return fail(DVBS2_EINVAL, "unsupported (standard, framesize, rate)");
    std::memset(out, 0, sizeof(*out));
    out->bch_k = fi.bch_k; out->bch_n = fi.bch_n; out->bch_t = fi.bch_t;
    out->ldpc_k = fi.ldpc_k; out->ldpc_n = fi.ldpc_n;
    if (fi.table) { out->table_k = fi.table->K; std::strncpy(out->table, fi.table->name, sizeof(out->table) - 1); }
    return DVBS2_OK;
}

const char* dvbs2_rate_name(int rate) { return rate_name(rate); }
int dvbs2_rate_from_name(const char* name)
{
    if (!name) return -1;
    for (int r = 0; r < num_rates(); r++) if (!std::strcmp(rate_name(r), name)) return r;
    return -1;
}

const char* dvbs2_ldpc_table_name(int index)
{
    const LdpcTableDesc* t = ldpc_table_at(index);
    return t ? t->name : nullptr;
}

int dvbs2_ldpc_table_info(const char* table, int* n, int* k, int* q, int* links_total, int* conflict_layers)
{
    API_TRY
    LdpcSchedule s;
    if (!compile_ldpc_schedule(find_ldpc_table(table), &s)) return fail(DVBS2_EINVAL, "unknown LDPC table");
    if (n) *n = s.N; if (k) *k = s.K; if (q) *q = s.q;
    if (links_total) *links_total = s.links_total;
    if (conflict_layers) *conflict_layers = s.conflict_layers;
    return DVBS2_OK;
    API_CATCH
}

int dvbs2_ldpc_layer_info(const char* table, int layer, int* block, int* groups, int* shifts, int max_entries)
{
    API_TRY
    LdpcSchedule s;
    if (!compile_ldpc_schedule(find_ldpc_table(table), &s)) return fail(DVBS2_EINVAL, "unknown LDPC table");
    if (layer < 0 || layer >= s.q) return fail(DVBS2_EINVAL, "layer out of range");
    const LdpcLayer& L = s.layers[layer];
    if (block) *block = L.block;
    for (int e = 0; e < L.cnt && e < max_entries; e++) {
        const LdpcEntry& en = s.entries[L.entry_off + e];
        if (groups) groups[e] = en.base / 360;
        if (shifts) shifts[e] = (360 - en.rot) % 360;
    }
    return L.cnt;
    API_CATCH
}

static int ldpc_make(dvbs2_ldpc_t** h, const LdpcTableDesc* t, int message_bits, int G, int max_frames, int device)
{
    if (!h) return fail(DVBS2_EINVAL, "null handle pointer");
    *h = nullptr;
    if (!t) return fail(DVBS2_EINVAL, "unknown LDPC table");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(DVBS2_EDEVICE, "no HIP device (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(DVBS2_EINVAL, "device index out of range");
    dvbs2_ldpc* o = new (std::nothrow) dvbs2_ldpc();
    if (!o) return fail(DVBS2_EDEVICE, "out of memory");
    o->device = device;
    if (const char* e = getenv("DVBS2_HOST_PLAN")) o->host_plan = e;
    if (const char* e = getenv("DVBS2_HOST_CHUNK")) o->host_chunk = std::max(2, atoi(e));
    if (const char* e = getenv("DVBS2_HOST_COPY_STREAM")) o->host_copy_stream = atoi(e) != 0 ? 1 : 0;
    o->dec = new (std::nothrow) LdpcDecoderHip(t, message_bits, G, max_frames, device);
    if (!o->dec || !o->dec->ok()) {
        std::string m = o->dec ? o->dec->error() : "out of memory";
        delete o->dec; delete o;
        return fail(m.find("hip") != std::string::npos ? DVBS2_EDEVICE : DVBS2_EINVAL, m);
    }
    *h = o;
    return DVBS2_OK;
}

int dvbs2_ldpc_create(dvbs2_ldpc_t** h, int standard, int framesize, int rate, int group_size, int max_frames, int device)
{
    API_TRY
    FecInfo fi;
    if (!get_fec_info(standard, framesize, rate, &fi) || !fi.table) return fail(DVBS2_EINVAL, "unsupported (standard, framesize, rate)");
    return ldpc_make(h, fi.table, (int)fi.ldpc_k, group_size, max_frames, device);
    API_CATCH
}

int dvbs2_ldpc_create_table(dvbs2_ldpc_t** h, const char* table, int message_bits, int group_size, int max_frames, int device)
{
    API_TRY
    return ldpc_make(h, find_ldpc_table(table), message_bits, group_size, max_frames, device);
    API_CATCH
}

void dvbs2_ldpc_destroy(dvbs2_ldpc_t* h)
{
    if (!h) return;
    DeviceGuard g(h->device);
    (void)hipFree(h->d_in); (void)hipFree(h->d_bits); (void)hipFree(h->d_llr); (void)hipFree(h->d_ret);
    if (h->p_bits) (void)hipHostFree(h->p_bits);
    if (h->p_llr) (void)hipHostFree(h->p_llr);
    if (h->p_ret) (void)hipHostFree(h->p_ret);
    for (hipStream_t st : h->stream) if (st) (void)hipStreamDestroy(st);
    if (h->copy_stream) (void)hipStreamDestroy(h->copy_stream);
    for (hipEvent_t ev : h->in_ready) if (ev) (void)hipEventDestroy(ev);
    delete h->dec;
    delete h;
}

int dvbs2_ldpc_params(const dvbs2_ldpc_t* h, int* n, int* table_k, int* message_bits, int* q, int* group_size)
{
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    if (n) *n = h->dec->N(); if (table_k) *table_k = h->dec->K();
    if (message_bits) *message_bits = h->dec->out_bits_message();
    if (q) *q = h->dec->q(); if (group_size) *group_size = h->dec->group_size();
    return DVBS2_OK;
}

static int ldpc_check_args(dvbs2_ldpc_t* h, const void* in, int n_frames, int max_trials, int out_mode, const void* bits,
                           const void* d_llr_out = nullptr, bool device_pointers = false)
{
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    if (n_frames < 0 || max_trials <= 0 || (n_frames && (!in || !bits))) return fail(DVBS2_EINVAL, "bad argument");
    // the kernels move LLRs with 8-byte loads and stores (include/dvbs2_fec_hip.h): a misaligned device pointer would be a GPU memory fault
    if (device_pointers && ((((uintptr_t)in) | ((uintptr_t)d_llr_out)) & 7u)) return fail(DVBS2_EINVAL, "d_llr_in / d_llr_out must be 8-byte aligned");
    if (out_mode != DVBS2_OM_CODEWORD && out_mode != DVBS2_OM_MESSAGE) return fail(DVBS2_EINVAL, "bad out_mode");
    if (n_frames > h->dec->max_frames()) return fail(DVBS2_ESIZE, "n_frames exceeds max_frames");
    return DVBS2_OK;
}

int dvbs2_ldpc_decode_device(dvbs2_ldpc_t* h, const int8_t* d_llr_in, int n_frames, int max_trials, int out_mode,
                             uint8_t* d_bits_out, int8_t* d_llr_out, int32_t* d_ret, void* stream)
{
    API_TRY
    if (int rc = ldpc_check_args(h, d_llr_in, n_frames, max_trials, out_mode, d_bits_out, d_llr_out, true)) return rc;
    if (h->dec->decode_device(d_llr_in, n_frames, max_trials, out_mode, d_bits_out, d_llr_out, d_ret, (hipStream_t)stream))
        return fail(DVBS2_EDEVICE, h->dec->error());
    return DVBS2_OK;
    API_CATCH
}

int dvbs2_ldpc_enqueue_device(dvbs2_ldpc_t* h, const int8_t* d_llr_in, int n_frames, int max_trials, int out_mode,
                              uint8_t* d_bits_out, int8_t* d_llr_out, int32_t* d_ret, void* stream)
{
    API_TRY
    if (int rc = ldpc_check_args(h, d_llr_in, n_frames, max_trials, out_mode, d_bits_out, d_llr_out, true)) return rc;
    if (h->dec->enqueue(d_llr_in, n_frames, max_trials, out_mode, d_bits_out, d_llr_out, d_ret, (hipStream_t)stream, 0, 0))
        return fail(DVBS2_EDEVICE, h->dec->error());
    return DVBS2_OK;
    API_CATCH
}

int dvbs2_ldpc_finish(dvbs2_ldpc_t* h)
{
    API_TRY
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    if (h->dec->finish(0) < 0) return fail(DVBS2_EDEVICE, h->dec->error());
    return DVBS2_OK;
    API_CATCH
}

int dvbs2_ldpc_decode(dvbs2_ldpc_t* h, const int8_t* llr_in, int n_frames, int max_trials, int out_mode,
                      uint8_t* bits_out, int8_t* llr_out, int32_t* ret)
{
    API_TRY
    if (int rc = ldpc_check_args(h, llr_in, n_frames, max_trials, out_mode, bits_out)) return rc;
    if (n_frames == 0) return DVBS2_OK;
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(DVBS2_EDEVICE, "hipSetDevice failed");
    const size_t N = h->dec->N(), mf = h->dec->max_frames();
    const int G = h->dec->group_size();
    for (hipStream_t& st : h->stream) if (!st) HCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    if (!h->copy_stream) HCHK(hipStreamCreateWithFlags(&h->copy_stream, hipStreamNonBlocking));
    for (hipEvent_t& ev : h->in_ready) if (!ev) HCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    if (!h->d_in) HCHK(hipMalloc(&h->d_in, mf * N));
    if (!h->d_bits) HCHK(hipMalloc(&h->d_bits, mf * (N / 8)));
    if (!h->d_llr) HCHK(hipMalloc(&h->d_llr, mf * N));
    if (!h->d_ret) HCHK(hipMalloc(&h->d_ret, ((mf + G - 1) / G + LdpcDecoderHip::kSlots) * 4));
    const size_t out_bytes = (out_mode ? h->dec->out_bits_message() : (int)N) / 8;
    // Results land where the DMA engine can write them: straight in the caller's buffer when it is page-locked (hipHostMalloc'ed, or
    // registered once with dvbs2_host_register -- what a block does with its item buffers), else in a pinned buffer of the handle
    // that is copied out when the chunk has finished.
    // (the WHOLE range has to lie inside one page-locked allocation / registration: host_range_page_locked)
    auto page_locked = [](const void* p, size_t bytes) { return host_range_page_locked(p, bytes); };
    const size_t n_groups = ((size_t)n_frames + G - 1) / G;
    uint8_t* bits_land = page_locked(bits_out, (size_t)n_frames * out_bytes) ? bits_out : nullptr;
    int8_t* llr_land = page_locked(llr_out, (size_t)n_frames * N) ? llr_out : nullptr;
    int32_t* ret_land = page_locked(ret, n_groups * 4) ? ret : nullptr;
    if (!bits_land) { if (!h->p_bits) HCHK(hipHostMalloc(&h->p_bits, mf * (N / 8))); bits_land = h->p_bits; }
    if (ret && !ret_land) { if (!h->p_ret) HCHK(hipHostMalloc(&h->p_ret, ((mf + G - 1) / G + LdpcDecoderHip::kSlots) * 4)); ret_land = h->p_ret; }
    if (llr_out && !llr_land) { if (!h->p_llr) HCHK(hipHostMalloc(&h->p_llr, mf * N)); llr_land = h->p_llr; }
    // The call is cut into chunks of whole groups (an even number of frames: two frames per workgroup); chunk c uses slot and stream
    // c % kSlots (its own range of the state and message buffers), so that transfers and decodes of neighbouring chunks overlap.
    int unit = G % 2 ? 2 * G : G;
    const bool in_locked = page_locked(llr_in, (size_t)n_frames * N);
    // Plan of the call, as (first frame, frames) chunks of whole groups. The decode of a chunk starts when its input has arrived and a
    // launch of up to 512 frames (one frame pair per CU) takes as long as a smaller one, so the FIRST chunk is 512 frames; every further
    // chunk boundary costs a launch gap that the single resident launch does not have. Pageable input: the runtime stages the copy
    // through its own pinned buffer while the calling thread waits, so the copy of chunk c + 1 has to run under the decode of chunk c:
    // chunks of 1024. Page-locked input (dvbs2_host_register / hipHostMalloc): the link moves 57 GB/s (bench.py host_link), the whole
    // input of 4096 normal frames arrives in under 5 ms through the copy stream: one large middle chunk, and a small last one so that
    // little output is left to fetch when the decode ends.
    std::vector<std::pair<int, int>> plan;
    auto round_unit = [&](int x) { return std::max(unit, (x + unit - 1) / unit * unit); };
    if (!h->host_plan.empty()) { // experiments: comma list of chunk sizes, the last one repeats
        int f0 = 0, last = 512;
        for (const char* q = h->host_plan.c_str(); f0 < n_frames;) {
            if (*q) { last = std::max(2, atoi(q)); while (*q && *q != ',') q++; if (*q == ',') q++; }
            const int nf = std::min(round_unit(last), n_frames - f0);
            plan.push_back({ f0, nf }); f0 += nf;
        }
    } else if (in_locked && n_frames > 1024 && !h->host_chunk) {
        // measured (MI355X, 4096 frames of table B4, tools/host_entry_ab.py): 512 | 3072 | 512 -> 97.3 % of the resident rate, 512 | 3584
        // 97.3 %, 512 | 1536 | 1536 | 512 96.7 %, eight chunks of 512 93.6 %, four of 1024 95.0 %
        // every boundary is a multiple of `unit` (frame_base of enqueue()) and none lies past the call's last frame (a group size above
        // 512 -- or an odd one above 256 -- makes `unit` larger than the first chunk: ADVICE r4)
        const int b1 = std::min(round_unit(512), n_frames);
        const int b2 = std::min(std::max(b1, (n_frames - 512) / unit * unit), n_frames);
        const int bounds[4] = { 0, b1, b2, n_frames };
        for (int k = 0; k < 3; k++) if (bounds[k + 1] > bounds[k]) plan.push_back({ bounds[k], bounds[k + 1] - bounds[k] });
    } else {
        // pageable input (measured as above): 512, then chunks of 1024 -> 96.0-96.5 %; eight equal chunks of 512 95.0-95.4 %
        int first = 512, chunk = std::max(1024, (n_frames + 7) / 8);
        if (h->host_chunk) first = chunk = h->host_chunk; // experiments, tests
        first = round_unit(first); chunk = round_unit(chunk);
        for (int f0 = 0; f0 < n_frames;) { const int nf = std::min(f0 ? chunk : first, n_frames - f0); plan.push_back({ f0, nf }); f0 += nf; }
    }
    const int n_chunks = (int)plan.size();
    bool use_copy_stream = in_locked;
    if (h->host_copy_stream >= 0) use_copy_stream = h->host_copy_stream != 0; // experiments
    auto copy_out = [&](int c) -> int {
        const int f0 = plan[c].first, nf = plan[c].second;
        hipStream_t st = h->stream[c % LdpcDecoderHip::kSlots];
        HCHK(hipMemcpyAsync(bits_land + (size_t)f0 * out_bytes, h->d_bits + (size_t)f0 * out_bytes, (size_t)nf * out_bytes, hipMemcpyDeviceToHost, st));
        if (llr_out) HCHK(hipMemcpyAsync(llr_land + (size_t)f0 * N, h->d_llr + (size_t)f0 * N, (size_t)nf * N, hipMemcpyDeviceToHost, st));
        if (ret) HCHK(hipMemcpyAsync(ret_land + f0 / G, h->d_ret + f0 / G, (size_t)((nf + G - 1) / G) * 4, hipMemcpyDeviceToHost, st));
        return DVBS2_OK;
    };
    auto finish = [&](int c) -> int {
        const int r = h->dec->finish(c % LdpcDecoderHip::kSlots);
        if (r < 0) return fail(DVBS2_EDEVICE, h->dec->error());
        if (r > 0) { if (int rc = copy_out(c)) return rc; } // outputs rewritten by the extra rounds: fetch them again
        HCHK(hipStreamSynchronize(h->stream[c % LdpcDecoderHip::kSlots]));
        const int f0 = plan[c].first, nf = plan[c].second;
        if (bits_land != bits_out) std::memcpy(bits_out + (size_t)f0 * out_bytes, bits_land + (size_t)f0 * out_bytes, (size_t)nf * out_bytes);
        if (llr_out && llr_land != llr_out) std::memcpy(llr_out + (size_t)f0 * N, llr_land + (size_t)f0 * N, (size_t)nf * N);
        if (ret && ret_land != ret) std::memcpy(ret + f0 / G, ret_land + f0 / G, (size_t)((nf + G - 1) / G) * 4);
        return DVBS2_OK;
    };
    // (a failure in the middle of the pipeline must not leave chunks in flight or slots busy: the handle stays usable)
    auto run = [&]() -> int {
        for (int c = 0; c < n_chunks; c++) {
            if (c >= LdpcDecoderHip::kSlots) if (int rc = finish(c - LdpcDecoderHip::kSlots)) return rc;
            const int f0 = plan[c].first, nf = plan[c].second;
            hipStream_t st = h->stream[c % LdpcDecoderHip::kSlots];
            if (use_copy_stream) {
                HCHK(hipMemcpyAsync(h->d_in + (size_t)f0 * N, llr_in + (size_t)f0 * N, (size_t)nf * N, hipMemcpyHostToDevice, h->copy_stream));
                HCHK(hipEventRecord(h->in_ready[c % LdpcDecoderHip::kSlots], h->copy_stream));
                HCHK(hipStreamWaitEvent(st, h->in_ready[c % LdpcDecoderHip::kSlots], 0));
            } else
                HCHK(hipMemcpyAsync(h->d_in + (size_t)f0 * N, llr_in + (size_t)f0 * N, (size_t)nf * N, hipMemcpyHostToDevice, st));
            if (h->dec->enqueue(h->d_in + (size_t)f0 * N, nf, max_trials, out_mode, h->d_bits + (size_t)f0 * out_bytes,
                                llr_out ? h->d_llr + (size_t)f0 * N : nullptr, h->d_ret + f0 / G, st, c % LdpcDecoderHip::kSlots, f0))
                return fail(DVBS2_EDEVICE, h->dec->error());
            if (int rc = copy_out(c)) return rc;
        }
        for (int c = std::max(0, n_chunks - LdpcDecoderHip::kSlots); c < n_chunks; c++) if (int rc = finish(c)) return rc;
        return DVBS2_OK;
    };
    const int rc = run();
    if (rc != DVBS2_OK) { // nothing of this call stays in flight (copies into the caller's buffers included)
        h->dec->abort_all();
        (void)hipStreamSynchronize(h->copy_stream);
        for (hipStream_t st : h->stream) (void)hipStreamSynchronize(st);
    }
    return rc;
    API_CATCH
}

const char* dvbs2_ldpc_kernel_name(const dvbs2_ldpc_t* h)
{
    return h ? h->dec->kernel_name() : nullptr;
}

int dvbs2_ldpc_fallback_rounds(const dvbs2_ldpc_t* h)
{
    return h ? h->dec->fallback_rounds() : -1;
}

int dvbs2_measure_host_copy(int device, size_t bytes, int n_streams, int kind, double* h2d_gbs, double* d2h_gbs)
{
    API_TRY
    if (!bytes || n_streams < 1 || n_streams > 16 || kind < 0 || kind > 2) return fail(DVBS2_EINVAL, "bad argument");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(DVBS2_EDEVICE, "hipSetDevice failed");
    void* host = nullptr; void* dev = nullptr;
    hipStream_t st[16] = {};
    hipEvent_t e0 = nullptr, e1 = nullptr, done[16] = {};
    int rc = DVBS2_OK;
    bool registered = false;
    auto body = [&]() -> int {
        if (kind == 0) HCHK(hipHostMalloc(&host, bytes));
        else if (kind == 1) {
            // a mapping of its own for the registration (whole pages nothing else lives in; shared anonymous memory: no copy-on-write, no
            // anonymous huge pages under it) -- see dvbs2_host_alloc in the header for why heap memory is not registered here any more
            host = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
            if (host == MAP_FAILED) { host = nullptr; return fail(DVBS2_EDEVICE, "out of host memory"); }
            std::memset(host, 1, bytes);
            HCHK(hipHostRegister(host, bytes, hipHostRegisterDefault));
            registered = true;
        } else {
            host = std::malloc(bytes);
            if (!host) return fail(DVBS2_EDEVICE, "out of host memory");
            std::memset(host, 1, bytes);
        }
        HCHK(hipMalloc(&dev, bytes));
        for (int i = 0; i < n_streams; i++) { HCHK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking)); HCHK(hipEventCreateWithFlags(&done[i], hipEventDisableTiming)); }
        HCHK(hipEventCreate(&e0)); HCHK(hipEventCreate(&e1));
        const size_t part = (bytes / n_streams) & ~(size_t)4095;
        for (int dir = 0; dir < 2; dir++) {
            double best = 0;
            for (int rep = 0; rep < 3; rep++) { // first repetition warms the path
                HCHK(hipDeviceSynchronize());
                HCHK(hipEventRecord(e0, st[0]));
                for (int i = 1; i < n_streams; i++) HCHK(hipStreamWaitEvent(st[i], e0, 0));
                for (int i = 0; i < n_streams; i++) {
                    char* hp = (char*)host + (size_t)i * part; char* dp = (char*)dev + (size_t)i * part;
                    if (dir == 0) HCHK(hipMemcpyAsync(dp, hp, part, hipMemcpyHostToDevice, st[i]));
                    else HCHK(hipMemcpyAsync(hp, dp, part, hipMemcpyDeviceToHost, st[i]));
                    if (i) { HCHK(hipEventRecord(done[i], st[i])); HCHK(hipStreamWaitEvent(st[0], done[i], 0)); }
                }
                HCHK(hipEventRecord(e1, st[0]));
                HCHK(hipEventSynchronize(e1));
                float ms = 0; HCHK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms > 0) best = std::max(best, (double)part * n_streams / (ms * 1e-3) / 1e9);
            }
            if (dir == 0) { if (h2d_gbs) *h2d_gbs = best; } else if (d2h_gbs) *d2h_gbs = best;
        }
        return DVBS2_OK;
    };
    rc = body();
    (void)hipDeviceSynchronize();
    for (int i = 0; i < n_streams; i++) { if (st[i]) (void)hipStreamDestroy(st[i]); if (done[i]) (void)hipEventDestroy(done[i]); }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (dev) (void)hipFree(dev);
    if (host) {
        if (kind == 0) (void)hipHostFree(host);
        else if (kind == 1) { if (registered) (void)hipHostUnregister(host); (void)munmap(host, bytes); }
        else std::free(host);
    }
    return rc;
    API_CATCH
}

} // extern "C"

// every SIMD of the device busy with dependent VALU adds; workgroup 0 reports its s_memtime (shader clock) and s_memrealtime (100 MHz) deltas
__global__ void dvbs2_clock_probe_kernel(unsigned long long* out, int n)
{
    unsigned long long r0, r1;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r0));
    const unsigned long long t0 = __builtin_readcyclecounter();
    uint32_t a = threadIdx.x;
    for (int i = 0; i < n; i++) asm volatile("v_add_u32 %0, %0, 1\n\tv_add_u32 %0, %0, 1\n\tv_add_u32 %0, %0, 1\n\tv_add_u32 %0, %0, 1" : "+v"(a));
    const unsigned long long t1 = __builtin_readcyclecounter();
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(r1));
    if (blockIdx.x == 0 && threadIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = a; }
}

extern "C" {

int dvbs2_measure_shader_clock(int device, double* ghz, double* kernel_ms)
{
    API_TRY
    if (!ghz) return fail(DVBS2_EINVAL, "bad argument");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(DVBS2_EDEVICE, "hipSetDevice failed");
    unsigned long long* d = nullptr; unsigned long long hv[3] = {};
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0;
    auto body = [&]() -> int {
        HCHK(hipMalloc(&d, 64));
        HCHK(hipEventCreate(&e0)); HCHK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; rep++) { // (the first launch lets the clock ramp)
            HCHK(hipEventRecord(e0, nullptr));
            hipLaunchKernelGGL(dvbs2_clock_probe_kernel, dim3(2048), dim3(256), 0, nullptr, d, 300000);
            HCHK(hipEventRecord(e1, nullptr));
            HCHK(hipEventSynchronize(e1));
        }
        HCHK(hipEventElapsedTime(&ms, e0, e1));
        HCHK(hipMemcpy(hv, d, 24, hipMemcpyDeviceToHost));
        return DVBS2_OK;
    };
    const int rc = body();
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (d) (void)hipFree(d);
    if (rc != DVBS2_OK) return rc;
    if (!hv[1]) return fail(DVBS2_EDEVICE, "clock probe returned nothing");
    *ghz = (double)hv[0] / (double)hv[1] * 0.1; // s_memrealtime counts at 100 MHz
    if (kernel_ms) *kernel_ms = ms;
    return DVBS2_OK;
    API_CATCH
}

int dvbs2_debug_cu_slot_table(int device, int device_key, unsigned long long* table_address, int* nonzero_words)
{
    API_TRY
    if (!table_address) return fail(DVBS2_EINVAL, "bad argument");
    DeviceGuard guard(device);
    if (!guard.ok) return fail(DVBS2_EDEVICE, "hipSetDevice failed");
    std::string e;
    int* t = cu_slot_table(device_key, &e);
    if (!t) return fail(DVBS2_EDEVICE, e);
    *table_address = (unsigned long long)(uintptr_t)t;
    if (nonzero_words) {
        std::vector<int> hv(kCuSlotWords);
        HCHK(hipMemcpy(hv.data(), t, hv.size() * 4, hipMemcpyDeviceToHost));
        int nz = 0;
        for (int v : hv) nz += v != 0;
        *nonzero_words = nz;
    }
    return DVBS2_OK;
    API_CATCH
}

int dvbs2_ldpc_profile(dvbs2_ldpc_t* h, int enable, double* total_ms, int* launches)
{
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    if (total_ms) *total_ms = h->dec->profile_ms();
    if (launches) *launches = h->dec->profile_launches();
    h->dec->set_profiling(enable != 0);
    if (enable) h->dec->reset_profile();
    return DVBS2_OK;
}

} // extern "C"

/* ------------------------------------------------------------------ BCH */
struct dvbs2_bch {
    BchDecoderHip* dec = nullptr;
    uint8_t* d_cw = nullptr; uint8_t* d_msg = nullptr; int32_t* d_corr = nullptr;
    hipStream_t stream = nullptr;
    int device = 0;
};

static int check_device(int device)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(DVBS2_EDEVICE, "no HIP device (this library has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(DVBS2_EINVAL, "device index out of range");
    return DVBS2_OK;
}

static int bch_make(dvbs2_bch_t** h, int m, uint32_t prim_poly, int t, int n, int max_frames, int device)
{
    if (!h) return fail(DVBS2_EINVAL, "null handle pointer");
    *h = nullptr;
    if (int rc = check_device(device)) return rc;
    dvbs2_bch* o = new (std::nothrow) dvbs2_bch();
    if (!o) return fail(DVBS2_EDEVICE, "out of memory");
    o->device = device;
    o->dec = new (std::nothrow) BchDecoderHip(m, prim_poly, t, n, max_frames, device);
    if (!o->dec || !o->dec->ok()) {
        std::string msg = o->dec ? o->dec->error() : "out of memory";
        delete o->dec; delete o;
        return fail(msg.find("hip") != std::string::npos ? DVBS2_EDEVICE : DVBS2_EINVAL, msg);
    }
    *h = o;
    return DVBS2_OK;
}

static void bch_field(int framesize, int* m, uint32_t* prim)
{   // reference lib/bch_decoder_bb_impl.cc:58-63
    if (framesize == DVBS2_FECFRAME_NORMAL) { *m = 16; *prim = 0x1002Du; }      // x^16 + x^5 + x^3 + x^2 + 1
    else if (framesize == DVBS2_FECFRAME_SHORT) { *m = 14; *prim = 0x402Bu; }   // x^14 + x^5 + x^3 + x + 1
    else { *m = 15; *prim = 0x802Du; }                                           // x^15 + x^5 + x^3 + x^2 + 1
}

extern "C" {

int dvbs2_bch_create(dvbs2_bch_t** h, int standard, int framesize, int rate, int max_frames, int device)
{
    API_TRY
    FecInfo fi;
    if (!get_fec_info(standard, framesize, rate, &fi)) return fail(DVBS2_EINVAL, "unsupported (standard, framesize, rate)");
    int m; uint32_t prim;
    bch_field(framesize, &m, &prim);
    int rc = bch_make(h, m, prim, (int)fi.bch_t, (int)fi.bch_n, max_frames, device);
    if (rc == DVBS2_OK && (*h)->dec->code().k != (int)fi.bch_k) { dvbs2_bch_destroy(*h); *h = nullptr; return fail(DVBS2_EINVAL, "BCH k mismatch with the parameter table"); }
    return rc;
    API_CATCH
}

int dvbs2_bch_create_raw(dvbs2_bch_t** h, int m, uint32_t prim_poly, int t, int n, int max_frames, int device)
{
    API_TRY
    return bch_make(h, m, prim_poly, t, n, max_frames, device);
    API_CATCH
}

void dvbs2_bch_destroy(dvbs2_bch_t* h)
{
    if (!h) return;
    DeviceGuard guard(h->device);
    (void)hipFree(h->d_cw); (void)hipFree(h->d_msg); (void)hipFree(h->d_corr);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h->dec;
    delete h;
}

int dvbs2_bch_params(const dvbs2_bch_t* h, int* n, int* k, int* t)
{
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    if (n) *n = h->dec->code().n; if (k) *k = h->dec->code().k; if (t) *t = h->dec->code().t;
    return DVBS2_OK;
}

int dvbs2_bch_genpoly(const dvbs2_bch_t* h, uint8_t* gen, int max_coefs)
{
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    const auto& g = h->dec->code().gen;
    if (gen) for (int i = 0; i < (int)g.size() && i < max_coefs; i++) gen[i] = g[i];
    return h->dec->code().gdeg;
}

int dvbs2_bch_set_descramble(dvbs2_bch_t* h, int enable)
{
    API_TRY
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    if (h->dec->set_descramble(enable != 0)) return fail(DVBS2_EDEVICE, h->dec->error());
    return DVBS2_OK;
    API_CATCH
}

int dvbs2_bb_descramble_sequence(uint8_t* seq, int n_bytes)
{
    if (!seq || n_bytes < 0 || n_bytes > 64800 / 8) return fail(DVBS2_EINVAL, "bad argument");
    bb_derandomise_sequence(seq, n_bytes);
    return DVBS2_OK;
}

int dvbs2_bch_decode_device(dvbs2_bch_t* h, const uint8_t* d_cw, int n_frames, uint8_t* d_msg, int32_t* d_corr, void* stream)
{
    API_TRY
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    if (n_frames < 0 || (n_frames && (!d_cw || !d_msg || !d_corr))) return fail(DVBS2_EINVAL, "bad argument");
    if (n_frames > h->dec->max_frames()) return fail(DVBS2_ESIZE, "n_frames exceeds max_frames");
    if (h->dec->decode_device(d_cw, n_frames, d_msg, d_corr, (hipStream_t)stream)) return fail(DVBS2_EDEVICE, h->dec->error());
    return DVBS2_OK;
    API_CATCH
}

int dvbs2_bch_decode(dvbs2_bch_t* h, const uint8_t* cw, int n_frames, uint8_t* msg, int32_t* corrections)
{
    API_TRY
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    if (n_frames < 0 || (n_frames && (!cw || !msg || !corrections))) return fail(DVBS2_EINVAL, "bad argument");
    if (n_frames > h->dec->max_frames()) return fail(DVBS2_ESIZE, "n_frames exceeds max_frames");
    if (n_frames == 0) return DVBS2_OK;
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(DVBS2_EDEVICE, "hipSetDevice failed");
    const size_t nb = h->dec->code().n / 8, kb = h->dec->code().k / 8, mf = h->dec->max_frames();
    if (!h->stream) HCHK(hipStreamCreate(&h->stream));
    if (!h->d_cw) HCHK(hipMalloc(&h->d_cw, mf * nb));
    if (!h->d_msg) HCHK(hipMalloc(&h->d_msg, mf * kb));
    if (!h->d_corr) HCHK(hipMalloc(&h->d_corr, mf * 4));
    HCHK(hipMemcpyAsync(h->d_cw, cw, (size_t)n_frames * nb, hipMemcpyHostToDevice, h->stream));
    if (h->dec->decode_device(h->d_cw, n_frames, h->d_msg, h->d_corr, h->stream)) return fail(DVBS2_EDEVICE, h->dec->error());
    HCHK(hipMemcpyAsync(msg, h->d_msg, (size_t)n_frames * kb, hipMemcpyDeviceToHost, h->stream));
    HCHK(hipMemcpyAsync(corrections, h->d_corr, (size_t)n_frames * 4, hipMemcpyDeviceToHost, h->stream));
    HCHK(hipStreamSynchronize(h->stream));
    return DVBS2_OK;
    API_CATCH
}

} // extern "C"

/* ------------------------------------------------------------------ demapper */
struct dvbs2_demap {
    DemapperHip* dm = nullptr;
    float* d_syms = nullptr; float* d_n0 = nullptr; int8_t* d_llr = nullptr; float* d_snr = nullptr;
    hipStream_t stream = nullptr;
    int device = 0;
};

extern "C" {

int dvbs2_demap_create(dvbs2_demap_t** h, int framesize, int rate, int constellation, int max_frames, int device)
{
    API_TRY
    if (!h) return fail(DVBS2_EINVAL, "null handle pointer");
    *h = nullptr;
    if (int rc = check_device(device)) return rc;
    dvbs2_demap* o = new (std::nothrow) dvbs2_demap();
    if (!o) return fail(DVBS2_EDEVICE, "out of memory");
    o->device = device;
    o->dm = new (std::nothrow) DemapperHip(framesize, rate, constellation, max_frames, device);
    if (!o->dm || !o->dm->ok()) { std::string msg = o->dm ? o->dm->error() : "out of memory"; delete o->dm; delete o; return fail(DVBS2_EINVAL, msg); }
    *h = o;
    return DVBS2_OK;
    API_CATCH
}

void dvbs2_demap_destroy(dvbs2_demap_t* h)
{
    if (!h) return;
    DeviceGuard guard(h->device);
    (void)hipFree(h->d_syms); (void)hipFree(h->d_n0); (void)hipFree(h->d_llr); (void)hipFree(h->d_snr);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h->dm;
    delete h;
}

int dvbs2_demap_params(const dvbs2_demap_t* h, int* n_syms, int* n_llr, int* n_mod, int* column_order)
{
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    if (n_syms) *n_syms = h->dm->n_syms(); if (n_llr) *n_llr = h->dm->n_llr();
    if (n_mod) *n_mod = h->dm->n_mod(); if (column_order) *column_order = h->dm->column_order();
    return DVBS2_OK;
}

int dvbs2_demap_soft_device(dvbs2_demap_t* h, const float* d_syms, int n_frames, const float* d_n0, int n0_count, int8_t* d_llr_out, void* stream)
{
    API_TRY
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    if (n_frames < 0 || (n_frames && (!d_syms || !d_n0 || !d_llr_out)) || (n0_count != 1 && n0_count != n_frames)) return fail(DVBS2_EINVAL, "bad argument");
    if (n_frames > h->dm->max_frames()) return fail(DVBS2_ESIZE, "n_frames exceeds max_frames");
    if (h->dm->soft_device(d_syms, n_frames, d_n0, n0_count, d_llr_out, (hipStream_t)stream)) return fail(DVBS2_EDEVICE, h->dm->error());
    return DVBS2_OK;
    API_CATCH
}

static int demap_stage(dvbs2_demap_t* h, const float* syms, int n_frames)
{
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(DVBS2_EDEVICE, "hipSetDevice failed");
    const size_t mf = h->dm->max_frames(), ns = h->dm->n_syms();
    if (!h->stream) HCHK(hipStreamCreate(&h->stream));
    if (!h->d_syms) HCHK(hipMalloc(&h->d_syms, mf * ns * 8));
    if (!h->d_n0) HCHK(hipMalloc(&h->d_n0, mf * 4));
    if (!h->d_llr) HCHK(hipMalloc(&h->d_llr, mf * h->dm->n_llr()));
    if (!h->d_snr) HCHK(hipMalloc(&h->d_snr, mf * 4));
    HCHK(hipMemcpyAsync(h->d_syms, syms, (size_t)n_frames * ns * 8, hipMemcpyHostToDevice, h->stream));
    return DVBS2_OK;
}

int dvbs2_demap_soft(dvbs2_demap_t* h, const float* syms, int n_frames, const float* n0, int n0_count, int8_t* llr_out)
{
    API_TRY
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    if (n_frames < 0 || (n_frames && (!syms || !n0 || !llr_out)) || (n0_count != 1 && n0_count != n_frames)) return fail(DVBS2_EINVAL, "bad argument");
    if (n_frames > h->dm->max_frames()) return fail(DVBS2_ESIZE, "n_frames exceeds max_frames");
    if (n_frames == 0) return DVBS2_OK;
    if (int rc = demap_stage(h, syms, n_frames)) return rc;
    HCHK(hipMemcpyAsync(h->d_n0, n0, (size_t)n0_count * 4, hipMemcpyHostToDevice, h->stream));
    if (h->dm->soft_device(h->d_syms, n_frames, h->d_n0, n0_count, h->d_llr, h->stream)) return fail(DVBS2_EDEVICE, h->dm->error());
    HCHK(hipMemcpyAsync(llr_out, h->d_llr, (size_t)n_frames * h->dm->n_llr(), hipMemcpyDeviceToHost, h->stream));
    HCHK(hipStreamSynchronize(h->stream));
    return DVBS2_OK;
    API_CATCH
}

int dvbs2_demap_estimate_snr_device(dvbs2_demap_t* h, const float* d_syms, int n_frames, float* d_snr_lin, void* stream)
{
    API_TRY
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    if (n_frames < 0 || (n_frames && (!d_syms || !d_snr_lin))) return fail(DVBS2_EINVAL, "bad argument");
    if (n_frames > h->dm->max_frames()) return fail(DVBS2_ESIZE, "n_frames exceeds max_frames");
    if (h->dm->snr_device(d_syms, nullptr, n_frames, d_snr_lin, (hipStream_t)stream)) return fail(DVBS2_EDEVICE, h->dm->error());
    return DVBS2_OK;
    API_CATCH
}

int dvbs2_demap_estimate_snr(dvbs2_demap_t* h, const float* syms, int n_frames, float* snr_lin)
{
    API_TRY
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    if (n_frames < 0 || (n_frames && (!syms || !snr_lin))) return fail(DVBS2_EINVAL, "bad argument");
    if (n_frames > h->dm->max_frames()) return fail(DVBS2_ESIZE, "n_frames exceeds max_frames");
    if (n_frames == 0) return DVBS2_OK;
    if (int rc = demap_stage(h, syms, n_frames)) return rc;
    if (h->dm->snr_device(h->d_syms, nullptr, n_frames, h->d_snr, h->stream)) return fail(DVBS2_EDEVICE, h->dm->error());
    HCHK(hipMemcpyAsync(snr_lin, h->d_snr, (size_t)n_frames * 4, hipMemcpyDeviceToHost, h->stream));
    HCHK(hipStreamSynchronize(h->stream));
    return DVBS2_OK;
    API_CATCH
}

int dvbs2_demap_refine_snr_device(dvbs2_demap_t* h, const float* d_syms, const int8_t* d_ref_llr, int n_frames, float* d_snr_lin, void* stream)
{
    API_TRY
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    if (n_frames < 0 || (n_frames && (!d_syms || !d_ref_llr || !d_snr_lin))) return fail(DVBS2_EINVAL, "bad argument");
    if (n_frames > h->dm->max_frames()) return fail(DVBS2_ESIZE, "n_frames exceeds max_frames");
    if (h->dm->snr_device(d_syms, d_ref_llr, n_frames, d_snr_lin, (hipStream_t)stream)) return fail(DVBS2_EDEVICE, h->dm->error());
    return DVBS2_OK;
    API_CATCH
}

int dvbs2_demap_refine_snr(dvbs2_demap_t* h, const float* syms, const int8_t* ref_llr, int n_frames, float* snr_lin)
{
    API_TRY
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    if (n_frames < 0 || (n_frames && (!syms || !ref_llr || !snr_lin))) return fail(DVBS2_EINVAL, "bad argument");
    if (n_frames > h->dm->max_frames()) return fail(DVBS2_ESIZE, "n_frames exceeds max_frames");
    if (n_frames == 0) return DVBS2_OK;
    if (int rc = demap_stage(h, syms, n_frames)) return rc;
    HCHK(hipMemcpyAsync(h->d_llr, ref_llr, (size_t)n_frames * h->dm->n_llr(), hipMemcpyHostToDevice, h->stream));
    if (h->dm->snr_device(h->d_syms, h->d_llr, n_frames, h->d_snr, h->stream)) return fail(DVBS2_EDEVICE, h->dm->error());
    HCHK(hipMemcpyAsync(snr_lin, h->d_snr, (size_t)n_frames * 4, hipMemcpyDeviceToHost, h->stream));
    HCHK(hipStreamSynchronize(h->stream));
    return DVBS2_OK;
    API_CATCH
}

} // extern "C"

/* ------------------------------------------------------------------ chain */
struct dvbs2_chain {
    dvbs2_demap_t* dm = nullptr; // absent for an LLR-domain chain (dvbs2_chain_create_llr)
    dvbs2_ldpc_t* ldpc = nullptr; dvbs2_bch_t* bch = nullptr;
    int8_t* d_llr = nullptr; uint8_t* d_bits = nullptr; int32_t* d_corr = nullptr;
    int device = 0, max_frames = 0, n_llr = 0, ldpc_bytes = 0, msg_bytes = 0;
    // the call between enqueue and finish
    bool pending = false; int n_frames = 0; uint8_t* d_msg = nullptr; int32_t* d_bch_corr = nullptr; void* stream = nullptr;
    // host-pointer entries (dvbs2_chain_decode / dvbs2_chain_decode_llr): device copies of the caller's buffers, pinned landing buffers for
    // the outputs of a pageable caller, one stream per chunk slot + one copy stream (the plan of dvbs2_ldpc_decode)
    float* hd_syms = nullptr; int8_t* hd_llr = nullptr; float* hd_n0 = nullptr; uint8_t* hd_msg = nullptr; int32_t* hd_ret = nullptr; int32_t* hd_corr = nullptr;
    uint8_t* p_msg = nullptr; int32_t* p_ret = nullptr; int32_t* p_corr = nullptr;
    hipStream_t hstream[LdpcDecoderHip::kSlots] = {};
    hipStream_t hcopy = nullptr;
    hipEvent_t h_in_ready[LdpcDecoderHip::kSlots] = {};
    int host_chunk = 0; // DVBS2_HOST_CHUNK (experiments, tests), read once at create
};

static int chain_make(dvbs2_chain_t** h, int standard, int framesize, int rate, int constellation, bool with_demap,
                      int group_size, int max_frames, int device)
{
    if (!h) return fail(DVBS2_EINVAL, "null handle pointer");
    *h = nullptr;
    dvbs2_chain* o = new (std::nothrow) dvbs2_chain();
    if (!o) return fail(DVBS2_EDEVICE, "out of memory");
    o->device = device; o->max_frames = max_frames;
    if (const char* e = getenv("DVBS2_HOST_CHUNK")) o->host_chunk = std::max(2, atoi(e));
    int rc = DVBS2_OK;
    if (with_demap) rc = dvbs2_demap_create(&o->dm, framesize, rate, constellation, max_frames, device);
    if (rc == DVBS2_OK) rc = dvbs2_ldpc_create(&o->ldpc, standard, framesize, rate, group_size, max_frames, device);
    if (rc == DVBS2_OK) rc = dvbs2_bch_create(&o->bch, standard, framesize, rate, max_frames, device);
    if (rc != DVBS2_OK) { std::string keep = g_err; dvbs2_chain_destroy(o); return fail(rc, keep); }
    o->n_llr = o->ldpc->dec->N();
    o->ldpc_bytes = o->ldpc->dec->out_bits_message() / 8;
    o->msg_bytes = o->bch->dec->code().k / 8;
    if ((o->dm && o->dm->dm->n_llr() != o->n_llr) || o->ldpc_bytes != o->bch->dec->code().n / 8) { dvbs2_chain_destroy(o); return fail(DVBS2_EINVAL, "inconsistent chain sizes"); }
    DeviceGuard guard(device);
    hipError_t e = guard.ok ? hipSuccess : hipErrorInvalidDevice;
    if (e == hipSuccess && o->dm) e = hipMalloc(&o->d_llr, (size_t)max_frames * o->n_llr);
    if (e == hipSuccess) e = hipMalloc(&o->d_bits, (size_t)max_frames * o->ldpc_bytes);
    if (e == hipSuccess) e = hipMalloc(&o->d_corr, (size_t)max_frames * 4);
    if (e != hipSuccess) { dvbs2_chain_destroy(o); return fail(DVBS2_EDEVICE, hipGetErrorString(e)); }
    *h = o;
    return DVBS2_OK;
}

// LDPC (already enqueued) -> BCH on the same stream; the LDPC output never leaves HBM
// BCH straight from the LDPC decoder's state (hard decision + packing of ldpc_decoder_bb fused into the BCH kernel's load: no
// finalize launch, no packed-bit buffer in between)
static int chain_bch_range(dvbs2_chain_t* h, int frame_base, int n_frames, uint8_t* d_msg, int32_t* d_corr, hipStream_t stream)
{
    const int N = h->ldpc->dec->N();
    if (h->bch->dec->decode_device(nullptr, n_frames, d_msg, d_corr, stream, h->ldpc->dec->state() + (size_t)frame_base * N, N, frame_base))
        return fail(DVBS2_EDEVICE, h->bch->dec->error());
    return DVBS2_OK;
}
static int chain_bch(dvbs2_chain_t* h) { return chain_bch_range(h, 0, h->n_frames, h->d_msg, h->d_bch_corr, (hipStream_t)h->stream); }

// LDPC -> BCH on one stream; dm != nullptr: the LDPC sweep kernel demaps the symbols while it loads them
static int chain_enqueue_tail(dvbs2_chain_t* h, const int8_t* d_llr, const DemapFused* dm, int n_frames, int max_trials, uint8_t* d_msg,
                              int32_t* d_ldpc_ret, int32_t* d_bch_corr, void* stream)
{
    if (h->ldpc->dec->enqueue(d_llr, n_frames, max_trials, DVBS2_OM_MESSAGE, nullptr, nullptr, d_ldpc_ret, (hipStream_t)stream, 0, 0, dm))
        return fail(DVBS2_EDEVICE, h->ldpc->dec->error());
    h->pending = true; h->n_frames = n_frames; h->d_msg = d_msg; h->d_bch_corr = d_bch_corr ? d_bch_corr : h->d_corr; h->stream = stream;
    const int rc = chain_bch(h);
    if (rc != DVBS2_OK) { h->ldpc->dec->abort_all(); h->pending = false; } // nothing stays in flight or busy after a failed call
    return rc;
}

extern "C" {

int dvbs2_chain_set_descramble(dvbs2_chain_t* h, int enable)
{
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    return dvbs2_bch_set_descramble(h->bch, enable);
}

void dvbs2_chain_destroy(dvbs2_chain_t* h)
{
    if (!h) return;
    DeviceGuard guard(h->device);
    dvbs2_demap_destroy(h->dm); dvbs2_ldpc_destroy(h->ldpc); dvbs2_bch_destroy(h->bch);
    (void)hipFree(h->d_llr); (void)hipFree(h->d_bits); (void)hipFree(h->d_corr);
    (void)hipFree(h->hd_syms); (void)hipFree(h->hd_llr); (void)hipFree(h->hd_n0); (void)hipFree(h->hd_msg); (void)hipFree(h->hd_ret); (void)hipFree(h->hd_corr);
    if (h->p_msg) (void)hipHostFree(h->p_msg);
    if (h->p_ret) (void)hipHostFree(h->p_ret);
    if (h->p_corr) (void)hipHostFree(h->p_corr);
    for (hipStream_t st : h->hstream) if (st) (void)hipStreamDestroy(st);
    if (h->hcopy) (void)hipStreamDestroy(h->hcopy);
    for (hipEvent_t ev : h->h_in_ready) if (ev) (void)hipEventDestroy(ev);
    delete h;
}

int dvbs2_chain_create(dvbs2_chain_t** h, int standard, int framesize, int rate, int constellation, int group_size, int max_frames, int device)
{
    API_TRY
    return chain_make(h, standard, framesize, rate, constellation, true, group_size, max_frames, device);
    API_CATCH
}

int dvbs2_chain_create_llr(dvbs2_chain_t** h, int standard, int framesize, int rate, int group_size, int max_frames, int device)
{
    API_TRY
    return chain_make(h, standard, framesize, rate, 0, false, group_size, max_frames, device);
    API_CATCH
}

int dvbs2_chain_params(const dvbs2_chain_t* h, int* n_syms, int* msg_bytes)
{
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    if (n_syms) *n_syms = h->dm ? h->dm->dm->n_syms() : 0;
    if (msg_bytes) *msg_bytes = h->msg_bytes;
    return DVBS2_OK;
}

int dvbs2_chain_llr_params(const dvbs2_chain_t* h, int* n_llr, int* msg_bytes, int* group_size)
{
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    if (n_llr) *n_llr = h->n_llr;
    if (msg_bytes) *msg_bytes = h->msg_bytes;
    if (group_size) *group_size = h->ldpc->dec->group_size();
    return DVBS2_OK;
}

int dvbs2_chain_enqueue_device(dvbs2_chain_t* h, const float* d_syms, int n_frames, const float* d_n0, int n0_count,
                               int max_trials, uint8_t* d_msg, int32_t* d_ldpc_ret, int32_t* d_bch_corr, void* stream)
{
    API_TRY
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    if (!h->dm) return fail(DVBS2_EINVAL, "this chain starts at LLRs: use dvbs2_chain_enqueue_llr_device");
    if (h->pending) return fail(DVBS2_EINVAL, "previous call not finished");
    if (n_frames > h->max_frames) return fail(DVBS2_ESIZE, "n_frames exceeds max_frames");
    if (n_frames < 0 || max_trials <= 0 || (n_frames && !d_msg)) return fail(DVBS2_EINVAL, "bad argument");
    if (n_frames == 0) return DVBS2_OK;
    if (!d_syms || !d_n0 || (n0_count != 1 && n0_count != n_frames)) return fail(DVBS2_EINVAL, "bad argument");
    if (h->ldpc->dec->fused_demap_supported()) { // symbols -> LDS inside the LDPC sweep kernel: no demapper launch, no LLR buffer
        const DemapFused dm = h->dm->dm->fused(d_syms, d_n0, n0_count);
        return chain_enqueue_tail(h, nullptr, &dm, n_frames, max_trials, d_msg, d_ldpc_ret, d_bch_corr, stream);
    }
    int rc = dvbs2_demap_soft_device(h->dm, d_syms, n_frames, d_n0, n0_count, h->d_llr, stream);
    if (rc == DVBS2_OK) rc = chain_enqueue_tail(h, h->d_llr, nullptr, n_frames, max_trials, d_msg, d_ldpc_ret, d_bch_corr, stream);
    return rc;
    API_CATCH
}

int dvbs2_chain_enqueue_llr_device(dvbs2_chain_t* h, const int8_t* d_llr, int n_frames, int max_trials, uint8_t* d_msg,
                                   int32_t* d_ldpc_ret, int32_t* d_bch_corr, void* stream)
{
    API_TRY
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    if (h->pending) return fail(DVBS2_EINVAL, "previous call not finished");
    if (n_frames > h->max_frames) return fail(DVBS2_ESIZE, "n_frames exceeds max_frames");
    if (n_frames < 0 || max_trials <= 0 || (n_frames && (!d_llr || !d_msg))) return fail(DVBS2_EINVAL, "bad argument");
    if (((uintptr_t)d_llr) & 7u) return fail(DVBS2_EINVAL, "d_llr must be 8-byte aligned"); // (8-byte loads: include/dvbs2_fec_hip.h)
    if (n_frames == 0) return DVBS2_OK;
    return chain_enqueue_tail(h, d_llr, nullptr, n_frames, max_trials, d_msg, d_ldpc_ret, d_bch_corr, stream);
    API_CATCH
}

int dvbs2_chain_ldpc_profile(dvbs2_chain_t* h, int enable, double* total_ms, int* launches)
{
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    return dvbs2_ldpc_profile(h->ldpc, enable, total_ms, launches);
}

const char* dvbs2_chain_ldpc_kernel_name(const dvbs2_chain_t* h) { return h ? h->ldpc->dec->kernel_name() : nullptr; }

int dvbs2_chain_finish(dvbs2_chain_t* h)
{
    API_TRY
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    if (!h->pending) return DVBS2_OK;
    h->pending = false;
    const int r = h->ldpc->dec->finish(0); // waits for the stream: demapper, LDPC and BCH of this call are done
    if (r < 0) return fail(DVBS2_EDEVICE, h->ldpc->dec->error());
    if (r > 0) { // the LDPC needed rounds beyond the enqueued ones and rewrote its output: run the BCH stage again
        int rc = chain_bch(h);
        if (rc != DVBS2_OK) return rc;
        HCHK(hipStreamSynchronize((hipStream_t)h->stream));
    }
    return DVBS2_OK;
    API_CATCH
}

int dvbs2_chain_decode_device(dvbs2_chain_t* h, const float* d_syms, int n_frames, const float* d_n0, int n0_count,
                              int max_trials, uint8_t* d_msg, int32_t* d_ldpc_ret, int32_t* d_bch_corr, void* stream)
{
    int rc = dvbs2_chain_enqueue_device(h, d_syms, n_frames, d_n0, n0_count, max_trials, d_msg, d_ldpc_ret, d_bch_corr, stream);
    if (rc != DVBS2_OK) { if (h) h->pending = false; return rc; }
    return dvbs2_chain_finish(h);
}

int dvbs2_chain_decode_llr_device(dvbs2_chain_t* h, const int8_t* d_llr, int n_frames, int max_trials, uint8_t* d_msg,
                                  int32_t* d_ldpc_ret, int32_t* d_bch_corr, void* stream)
{
    int rc = dvbs2_chain_enqueue_llr_device(h, d_llr, n_frames, max_trials, d_msg, d_ldpc_ret, d_bch_corr, stream);
    if (rc != DVBS2_OK) { if (h) h->pending = false; return rc; }
    return dvbs2_chain_finish(h);
}

} // extern "C"

// Host-pointer form of the fused chain (SURVEY 8(b) "dvbs2_fec_chain_decode(syms -> msg bytes)"): what the three blocks do with the
// item buffers GNU Radio hands them (lib/xfecframe_demapper_cb_impl.cc:101-186 -> lib/ldpc_decoder_bb_impl.cc:394-455 ->
// lib/bch_decoder_bb_impl.cc:84-117), as ONE call. The call is cut into chunks of whole LDPC groups exactly like dvbs2_ldpc_decode:
// chunk c runs on stream c mod kSlots in its own range of the LDPC state / message buffers and of the BCH syndrome words, its input
// copy (through one copy stream when the caller's buffer is page-locked) runs under the kernels of chunk c - 1 and its results go
// back while chunk c + 1 decodes. An 8PSK normal frame is 172.8 KB of symbols in and ~6 KB out: at the ~57 GB/s of the host link the
// chain is LINK-bound near 320 k frames/s -- below what the kernels do at a receiver's operating point (bench.py config3_host).
// in_syms: XFECFRAME symbols (a demapping chain) or null; in_llr: int8 LLRs (either kind of chain) or null.
static int chain_decode_host(dvbs2_chain_t* h, const float* in_syms, const int8_t* in_llr, int n_frames, const float* n0, int n0_count,
                             int max_trials, uint8_t* msg, int32_t* ldpc_ret, int32_t* bch_corr)
{
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    if (h->pending) return fail(DVBS2_EINVAL, "previous call not finished");
    if (n_frames > h->max_frames) return fail(DVBS2_ESIZE, "n_frames exceeds max_frames");
    if (n_frames < 0 || max_trials <= 0 || (n_frames && (!msg || (!in_syms && !in_llr)))) return fail(DVBS2_EINVAL, "bad argument");
    if (in_syms && !h->dm) return fail(DVBS2_EINVAL, "this chain starts at LLRs: use dvbs2_chain_decode_llr");
    if (in_syms && (!n0 || (n0_count != 1 && n0_count != n_frames))) return fail(DVBS2_EINVAL, "bad argument");
    if (n_frames == 0) return DVBS2_OK;
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(DVBS2_EDEVICE, "hipSetDevice failed");
    constexpr int kSl = LdpcDecoderHip::kSlots;
    LdpcDecoderHip* dec = h->ldpc->dec;
    const size_t N = (size_t)dec->N(), mf = (size_t)h->max_frames, mb = (size_t)h->msg_bytes;
    const size_t ns = h->dm ? (size_t)h->dm->dm->n_syms() : 0;
    const int G = dec->group_size();
    for (hipStream_t& st : h->hstream) if (!st) HCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    if (!h->hcopy) HCHK(hipStreamCreateWithFlags(&h->hcopy, hipStreamNonBlocking));
    for (hipEvent_t& ev : h->h_in_ready) if (!ev) HCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    const bool fused = in_syms && dec->fused_demap_supported(); // symbols -> LDS inside the sweep kernel; else demapper launch -> LLR buffer
    if (in_syms && !h->hd_syms) HCHK(hipMalloc(&h->hd_syms, mf * ns * 8));
    if (in_syms && !h->hd_n0) HCHK(hipMalloc(&h->hd_n0, mf * 4));
    if ((in_llr || !fused) && !h->hd_llr) HCHK(hipMalloc(&h->hd_llr, mf * N));
    if (!h->hd_msg) HCHK(hipMalloc(&h->hd_msg, mf * mb));
    if (!h->hd_ret) HCHK(hipMalloc(&h->hd_ret, ((mf + G - 1) / G + kSl) * 4));
    if (!h->hd_corr) HCHK(hipMalloc(&h->hd_corr, mf * 4));
    const size_t n_groups = ((size_t)n_frames + G - 1) / G;
    uint8_t* msg_land = host_range_page_locked(msg, (size_t)n_frames * mb) ? msg : nullptr;
    int32_t* ret_land = host_range_page_locked(ldpc_ret, n_groups * 4) ? ldpc_ret : nullptr;
    int32_t* corr_land = host_range_page_locked(bch_corr, (size_t)n_frames * 4) ? bch_corr : nullptr;
    if (!msg_land) { if (!h->p_msg) HCHK(hipHostMalloc(&h->p_msg, mf * mb)); msg_land = h->p_msg; }
    if (ldpc_ret && !ret_land) { if (!h->p_ret) HCHK(hipHostMalloc(&h->p_ret, ((mf + G - 1) / G + kSl) * 4)); ret_land = h->p_ret; }
    if (bch_corr && !corr_land) { if (!h->p_corr) HCHK(hipHostMalloc(&h->p_corr, mf * 4)); corr_land = h->p_corr; }
    const void* in_ptr = in_syms ? (const void*)in_syms : (const void*)in_llr;
    const size_t in_frame_bytes = in_syms ? ns * 8 : N;
    const bool in_locked = host_range_page_locked(in_ptr, (size_t)n_frames * in_frame_bytes);
    // chunk plan: as dvbs2_ldpc_decode (first chunk 512 frames = one launch wave of frame pairs; page-locked input: one large middle chunk
    // and a small last one; pageable input: chunks of 1024 so that the staged copy of chunk c + 1 runs under the decode of chunk c)
    const int unit = G % 2 ? 2 * G : G;
    auto round_unit = [&](int x) { return std::max(unit, (x + unit - 1) / unit * unit); };
    std::vector<std::pair<int, int>> plan;
    if (in_syms && in_locked && !h->host_chunk) {
        // SYMBOL input is 8 bytes per symbol (172.8 KB per 8PSK normal frame, 2.7 x the LLR bytes): at a receiver's operating point the copy of a
        // chunk takes longer than its kernels, so page-locked input goes in uniform chunks of 512 frames (one launch wave of the GPU) -- the copy of
        // chunk c + 1 runs under the kernels of chunk c and the call ends one chunk's kernels after its last byte arrived. Measured (MI355X, 4096
        // frames of 8PSK 3/4 normal, page-locked buffers): 512 | 3072 | 512 (the LDPC entry's plan) 221 k frames/s at Es/N0 8.5 dB = 0.67 of the link
        // bound -- the GPU idles while 531 MB arrive -- and 0.853 of the resident rate on never-converging input; uniform 512: 291 k (0.88 of the
        // link bound) and 0.899. Chunks of 128 for 512-frame calls LOSE (a launch of 128 frames takes as long as one of 512): 181 -> 144 k.
        const int chunk = round_unit(512);
        for (int f0 = 0; f0 < n_frames; f0 += chunk) plan.push_back({ f0, std::min(chunk, n_frames - f0) });
    } else if (in_locked && n_frames > 1024 && !h->host_chunk) {
        const int b1 = std::min(round_unit(512), n_frames);
        const int b2 = std::min(std::max(b1, (n_frames - 512) / unit * unit), n_frames);
        const int bounds[4] = { 0, b1, b2, n_frames };
        for (int k = 0; k < 3; k++) if (bounds[k + 1] > bounds[k]) plan.push_back({ bounds[k], bounds[k + 1] - bounds[k] });
    } else {
        int first = 512, chunk = std::max(1024, (n_frames + 7) / 8);
        if (h->host_chunk) first = chunk = h->host_chunk;
        first = round_unit(first); chunk = round_unit(chunk);
        for (int f0 = 0; f0 < n_frames;) { const int nf = std::min(f0 ? chunk : first, n_frames - f0); plan.push_back({ f0, nf }); f0 += nf; }
    }
    const int n_chunks = (int)plan.size();
    auto copy_out = [&](int c) -> int {
        const int f0 = plan[c].first, nf = plan[c].second;
        hipStream_t st = h->hstream[c % kSl];
        HCHK(hipMemcpyAsync(msg_land + (size_t)f0 * mb, h->hd_msg + (size_t)f0 * mb, (size_t)nf * mb, hipMemcpyDeviceToHost, st));
        if (ldpc_ret) HCHK(hipMemcpyAsync(ret_land + f0 / G, h->hd_ret + f0 / G, (size_t)((nf + G - 1) / G) * 4, hipMemcpyDeviceToHost, st));
        if (bch_corr) HCHK(hipMemcpyAsync(corr_land + f0, h->hd_corr + f0, (size_t)nf * 4, hipMemcpyDeviceToHost, st));
        return DVBS2_OK;
    };
    auto finish = [&](int c) -> int {
        const int f0 = plan[c].first, nf = plan[c].second;
        hipStream_t st = h->hstream[c % kSl];
        const int r = dec->finish(c % kSl);
        if (r < 0) return fail(DVBS2_EDEVICE, dec->error());
        if (r > 0) { // the LDPC needed rounds beyond the enqueued ones and rewrote its state: BCH and the copies again
            if (int rc = chain_bch_range(h, f0, nf, h->hd_msg + (size_t)f0 * mb, h->hd_corr + f0, st)) return rc;
            if (int rc = copy_out(c)) return rc;
        }
        HCHK(hipStreamSynchronize(st));
        if (msg_land != msg) std::memcpy(msg + (size_t)f0 * mb, msg_land + (size_t)f0 * mb, (size_t)nf * mb);
        if (ldpc_ret && ret_land != ldpc_ret) std::memcpy(ldpc_ret + f0 / G, ret_land + f0 / G, (size_t)((nf + G - 1) / G) * 4);
        if (bch_corr && corr_land != bch_corr) std::memcpy(bch_corr + f0, corr_land + f0, (size_t)nf * 4);
        return DVBS2_OK;
    };
    auto run = [&]() -> int {
        for (int c = 0; c < n_chunks; c++) {
            if (c >= kSl) if (int rc = finish(c - kSl)) return rc;
            const int f0 = plan[c].first, nf = plan[c].second;
            hipStream_t st = h->hstream[c % kSl];
            hipStream_t cs = in_locked ? h->hcopy : st; // (pageable input: the runtime stages the copy while the caller waits; its own stream)
            if (in_syms) {
                HCHK(hipMemcpyAsync(h->hd_syms + (size_t)f0 * ns * 2, in_syms + (size_t)f0 * ns * 2, (size_t)nf * ns * 8, hipMemcpyHostToDevice, cs));
                if (n0_count > 1) HCHK(hipMemcpyAsync(h->hd_n0 + f0, n0 + f0, (size_t)nf * 4, hipMemcpyHostToDevice, cs));
                else if (c == 0) HCHK(hipMemcpyAsync(h->hd_n0, n0, 4, hipMemcpyHostToDevice, cs));
            } else
                HCHK(hipMemcpyAsync(h->hd_llr + (size_t)f0 * N, in_llr + (size_t)f0 * N, (size_t)nf * N, hipMemcpyHostToDevice, cs));
            if (cs != st) {
                HCHK(hipEventRecord(h->h_in_ready[c % kSl], cs));
                HCHK(hipStreamWaitEvent(st, h->h_in_ready[c % kSl], 0));
            } else if (in_syms && n0_count == 1 && c > 0 && c < kSl) {
                // (the single N0 travelled on chunk 0's stream: the first chunks on the other streams wait for it)
                HCHK(hipStreamWaitEvent(st, h->h_in_ready[0], 0));
            }
            if (cs == st && in_syms && n0_count == 1 && c == 0) HCHK(hipEventRecord(h->h_in_ready[0], st));
            const float* dn0 = n0_count > 1 ? h->hd_n0 + f0 : h->hd_n0;
            const float* dsy = in_syms ? h->hd_syms + (size_t)f0 * ns * 2 : nullptr;
            int erc;
            if (fused) {
                const DemapFused dm = h->dm->dm->fused(dsy, dn0, n0_count > 1 ? nf : 1);
                erc = dec->enqueue(nullptr, nf, max_trials, DVBS2_OM_MESSAGE, nullptr, nullptr, h->hd_ret + f0 / G, st, c % kSl, f0, &dm);
            } else {
                if (in_syms && h->dm->dm->soft_device(dsy, nf, dn0, n0_count > 1 ? nf : 1, h->hd_llr + (size_t)f0 * N, st)) return fail(DVBS2_EDEVICE, h->dm->dm->error());
                erc = dec->enqueue(h->hd_llr + (size_t)f0 * N, nf, max_trials, DVBS2_OM_MESSAGE, nullptr, nullptr, h->hd_ret + f0 / G, st, c % kSl, f0, nullptr);
            }
            if (erc) return fail(DVBS2_EDEVICE, dec->error());
            if (int rc = chain_bch_range(h, f0, nf, h->hd_msg + (size_t)f0 * mb, h->hd_corr + f0, st)) return rc;
            if (int rc = copy_out(c)) return rc;
        }
        for (int c = std::max(0, n_chunks - kSl); c < n_chunks; c++) if (int rc = finish(c)) return rc;
        return DVBS2_OK;
    };
    const int rc = run();
    if (rc != DVBS2_OK) { // nothing of this call stays in flight (copies into the caller's buffers included)
        const std::string keep = g_err;
        dec->abort_all();
        (void)hipStreamSynchronize(h->hcopy);
        for (hipStream_t st : h->hstream) (void)hipStreamSynchronize(st);
        g_err = keep;
    }
    return rc;
}

extern "C" {

int dvbs2_chain_decode(dvbs2_chain_t* h, const float* syms, int n_frames, const float* n0, int n0_count, int max_trials,
                       uint8_t* msg, int32_t* ldpc_ret, int32_t* bch_corr)
{
    API_TRY
    if (n_frames > 0 && !syms) return fail(DVBS2_EINVAL, "bad argument");
    return chain_decode_host(h, syms, nullptr, n_frames, n0, n0_count, max_trials, msg, ldpc_ret, bch_corr);
    API_CATCH
}

int dvbs2_chain_decode_llr(dvbs2_chain_t* h, const int8_t* llr, int n_frames, int max_trials, uint8_t* msg, int32_t* ldpc_ret, int32_t* bch_corr)
{
    API_TRY
    if (n_frames > 0 && !llr) return fail(DVBS2_EINVAL, "bad argument");
    return chain_decode_host(h, nullptr, llr, n_frames, nullptr, 0, max_trials, msg, ldpc_ret, bch_corr);
    API_CATCH
}

} // extern "C"

/* ------------------------------------------------------------------ PLFRAME payload step */
struct dvbs2_plpayload {
    PlPayloadHip* pp = nullptr;
    float* d_in = nullptr; float* d_out = nullptr; float* d_par = nullptr; int32_t* d_cc = nullptr;
    hipStream_t stream = nullptr;
    int device = 0;
};

extern "C" {

int dvbs2_pl_scrambling_rn(int gold_code, uint8_t* rn, int n)
{
    API_TRY
    if (!rn || n < 0 || n > 360 * 90 + 22 * 36 || gold_code < 0 || gold_code >= (1 << 18) - 1) return fail(DVBS2_EINVAL, "bad argument");
    pl_scrambling_rn(gold_code, rn, n);
    return DVBS2_OK;
    API_CATCH
}

int dvbs2_plpayload_create(dvbs2_plpayload_t** h, int gold_code, int n_slots, int has_pilots, int max_frames, int device)
{
    API_TRY
    if (!h) return fail(DVBS2_EINVAL, "null handle pointer");
    *h = nullptr;
    if (int rc = check_device(device)) return rc;
    dvbs2_plpayload* o = new (std::nothrow) dvbs2_plpayload();
    if (!o) return fail(DVBS2_EDEVICE, "out of memory");
    o->device = device;
    o->pp = new (std::nothrow) PlPayloadHip(gold_code, n_slots, has_pilots, max_frames, device);
    if (!o->pp || !o->pp->ok()) { std::string msg = o->pp ? o->pp->error() : "out of memory"; delete o->pp; delete o; return fail(DVBS2_EINVAL, msg); }
    *h = o;
    return DVBS2_OK;
    API_CATCH
}

void dvbs2_plpayload_destroy(dvbs2_plpayload_t* h)
{
    if (!h) return;
    DeviceGuard guard(h->device);
    (void)hipFree(h->d_in); (void)hipFree(h->d_out); (void)hipFree(h->d_par); (void)hipFree(h->d_cc);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h->pp;
    delete h;
}

int dvbs2_plpayload_params(const dvbs2_plpayload_t* h, int* payload_len, int* xfecframe_len, int* n_pilots)
{
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    if (payload_len) *payload_len = h->pp->payload_len();
    if (xfecframe_len) *xfecframe_len = h->pp->xfecframe_len();
    if (n_pilots) *n_pilots = h->pp->n_pilots();
    return DVBS2_OK;
}

int dvbs2_plpayload_process_device(dvbs2_plpayload_t* h, const float* d_payload, int n_frames, const float* d_plheader_phase,
                                   const float* d_phase_inc, const int32_t* d_coarse_corrected, const float* d_pilot_phase,
                                   float* d_xfecframes, void* stream)
{
    API_TRY
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    if (n_frames < 0 || (n_frames && (!d_payload || !d_plheader_phase || !d_phase_inc || !d_coarse_corrected || !d_xfecframes ||
                                      (h->pp->n_pilots() > 0 && !d_pilot_phase)))) return fail(DVBS2_EINVAL, "bad argument");
    if (n_frames > h->pp->max_frames()) return fail(DVBS2_ESIZE, "n_frames exceeds max_frames");
    if (h->pp->process_device(d_payload, n_frames, d_plheader_phase, d_phase_inc, d_coarse_corrected, d_pilot_phase, d_xfecframes,
                              (hipStream_t)stream)) return fail(DVBS2_EDEVICE, h->pp->error());
    return DVBS2_OK;
    API_CATCH
}

int dvbs2_plpayload_process(dvbs2_plpayload_t* h, const float* payload, int n_frames, const float* plheader_phase, const float* phase_inc,
                            const int32_t* coarse_corrected, const float* pilot_phase, float* xfecframes)
{
    API_TRY
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    const int np = h->pp->n_pilots();
    if (n_frames < 0 || (n_frames && (!payload || !plheader_phase || !phase_inc || !coarse_corrected || !xfecframes || (np > 0 && !pilot_phase))))
        return fail(DVBS2_EINVAL, "bad argument");
    if (n_frames > h->pp->max_frames()) return fail(DVBS2_ESIZE, "n_frames exceeds max_frames");
    if (n_frames == 0) return DVBS2_OK;
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(DVBS2_EDEVICE, "hipSetDevice failed");
    const size_t mf = h->pp->max_frames(), pl = h->pp->payload_len(), xl = h->pp->xfecframe_len(), nf = n_frames;
    if (!h->stream) HCHK(hipStreamCreate(&h->stream));
    if (!h->d_in) HCHK(hipMalloc(&h->d_in, mf * pl * 8));
    if (!h->d_out) HCHK(hipMalloc(&h->d_out, mf * xl * 8));
    if (!h->d_par) HCHK(hipMalloc(&h->d_par, mf * (2 + (np ? np : 1)) * 4));
    if (!h->d_cc) HCHK(hipMalloc(&h->d_cc, mf * 4));
    float* d_hph = h->d_par; float* d_inc = h->d_par + mf; float* d_pp = h->d_par + 2 * mf;
    HCHK(hipMemcpyAsync(h->d_in, payload, nf * pl * 8, hipMemcpyHostToDevice, h->stream));
    HCHK(hipMemcpyAsync(d_hph, plheader_phase, nf * 4, hipMemcpyHostToDevice, h->stream));
    HCHK(hipMemcpyAsync(d_inc, phase_inc, nf * 4, hipMemcpyHostToDevice, h->stream));
    HCHK(hipMemcpyAsync(h->d_cc, coarse_corrected, nf * 4, hipMemcpyHostToDevice, h->stream));
    if (np) HCHK(hipMemcpyAsync(d_pp, pilot_phase, nf * np * 4, hipMemcpyHostToDevice, h->stream));
    if (h->pp->process_device(h->d_in, n_frames, d_hph, d_inc, h->d_cc, d_pp, h->d_out, h->stream)) return fail(DVBS2_EDEVICE, h->pp->error());
    HCHK(hipMemcpyAsync(xfecframes, h->d_out, nf * xl * 8, hipMemcpyDeviceToHost, h->stream));
    HCHK(hipStreamSynchronize(h->stream));
    return DVBS2_OK;
    API_CATCH
}

} // extern "C"

/* ------------------------------------------------------------------ BBFRAME de-header */
struct dvbs2_bbdeheader {
    BbDeheaderHip* bb = nullptr;
    uint8_t* d_in = nullptr; uint8_t* d_out = nullptr; // staging of the host entry
    hipStream_t stream = nullptr;
    int device = 0;
};

extern "C" {

int dvbs2_bbdeheader_create_raw(dvbs2_bbdeheader_t** h, int kbch_bits, int max_frames, int device)
{
    API_TRY
    if (!h) return fail(DVBS2_EINVAL, "null handle pointer");
    *h = nullptr;
    if (int rc = check_device(device)) return rc;
    dvbs2_bbdeheader* o = new (std::nothrow) dvbs2_bbdeheader();
    if (!o) return fail(DVBS2_EDEVICE, "out of memory");
    o->device = device;
    o->bb = new (std::nothrow) BbDeheaderHip(kbch_bits, max_frames, device);
    if (!o->bb || !o->bb->ok()) { std::string msg = o->bb ? o->bb->error() : "out of memory"; delete o->bb; delete o; return fail(DVBS2_EINVAL, msg); }
    *h = o;
    return DVBS2_OK;
    API_CATCH
}

int dvbs2_bbdeheader_create(dvbs2_bbdeheader_t** h, int standard, int framesize, int rate, int max_frames, int device)
{
    API_TRY
    if (!h) return fail(DVBS2_EINVAL, "null handle pointer");
    *h = nullptr;
    FecInfo fi;
    if (!get_fec_info(standard, framesize, rate, &fi)) return fail(DVBS2_EINVAL, "unsupported (standard, framesize, rate)");
    return dvbs2_bbdeheader_create_raw(h, (int)fi.bch_k, max_frames, device); // d_kbch_bytes, d_max_dfl (:58-61)
    API_CATCH
}

void dvbs2_bbdeheader_destroy(dvbs2_bbdeheader_t* h)
{
    if (!h) return;
    DeviceGuard guard(h->device);
    (void)hipFree(h->d_in); (void)hipFree(h->d_out);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h->bb;
    delete h;
}

int dvbs2_bbdeheader_params(const dvbs2_bbdeheader_t* h, int* kbch_bytes, int* max_dfl_bits, int* max_out_bytes_per_frame)
{
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    if (kbch_bytes) *kbch_bytes = h->bb->kbch_bytes();
    if (max_dfl_bits) *max_dfl_bits = h->bb->max_dfl();
    if (max_out_bytes_per_frame) *max_out_bytes_per_frame = h->bb->max_out_bytes_per_frame();
    return DVBS2_OK;
}

int dvbs2_bbdeheader_process_device(dvbs2_bbdeheader_t* h, const uint8_t* d_bbframes, int n_frames, uint8_t* d_ts_out, void* stream)
{
    API_TRY
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    if (n_frames < 0 || (n_frames && (!d_bbframes || !d_ts_out))) return fail(DVBS2_EINVAL, "bad argument");
    if (n_frames > h->bb->max_frames()) return fail(DVBS2_ESIZE, "n_frames exceeds max_frames");
    if (h->bb->process_device(d_bbframes, n_frames, d_ts_out, (hipStream_t)stream)) return fail(DVBS2_EDEVICE, h->bb->error());
    return DVBS2_OK;
    API_CATCH
}

int dvbs2_bbdeheader_finish(dvbs2_bbdeheader_t* h, int64_t* produced, void* stream)
{
    API_TRY
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    BbdhState st;
    if (h->bb->state(&st, (hipStream_t)stream)) return fail(DVBS2_EDEVICE, h->bb->error());
    if (produced) *produced = (int64_t)st.produced;
    return DVBS2_OK;
    API_CATCH
}

int dvbs2_bbdeheader_counters(dvbs2_bbdeheader_t* h, dvbs2_bbdeheader_counters_t* out, void* stream)
{
    API_TRY
    if (!h || !out) return fail(DVBS2_EINVAL, "bad argument");
    BbdhState st;
    if (h->bb->state(&st, (hipStream_t)stream)) return fail(DVBS2_EDEVICE, h->bb->error());
    out->packets = st.packets; out->errors = st.errors; out->bbframes = st.bbframes; out->dropped = st.dropped; out->gaps = st.gaps;
    out->overruns = st.overruns; out->synched = st.synched; out->partial_ts_bytes = st.partial;
    return DVBS2_OK;
    API_CATCH
}

int dvbs2_bbdeheader_reset(dvbs2_bbdeheader_t* h, void* stream)
{
    API_TRY
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    if (h->bb->reset((hipStream_t)stream)) return fail(DVBS2_EDEVICE, h->bb->error());
    return DVBS2_OK;
    API_CATCH
}

int dvbs2_bbdeheader_process(dvbs2_bbdeheader_t* h, const uint8_t* bbframes, int n_frames, uint8_t* ts_out, int64_t* produced)
{
    API_TRY
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    if (n_frames < 0 || (n_frames && (!bbframes || !ts_out))) return fail(DVBS2_EINVAL, "bad argument");
    if (n_frames > h->bb->max_frames()) return fail(DVBS2_ESIZE, "n_frames exceeds max_frames");
    if (produced) *produced = 0;
    DeviceGuard guard(h->device);
    if (!guard.ok) return fail(DVBS2_EDEVICE, "hipSetDevice failed");
    const size_t mf = h->bb->max_frames(), fb = h->bb->kbch_bytes(), ob = h->bb->max_out_bytes_per_frame();
    if (!h->stream) HCHK(hipStreamCreate(&h->stream));
    if (!h->d_in) HCHK(hipMalloc(&h->d_in, mf * fb));
    if (!h->d_out) HCHK(hipMalloc(&h->d_out, mf * ob));
    if (n_frames) HCHK(hipMemcpyAsync(h->d_in, bbframes, (size_t)n_frames * fb, hipMemcpyHostToDevice, h->stream));
    if (h->bb->process_device(h->d_in, n_frames, h->d_out, h->stream)) return fail(DVBS2_EDEVICE, h->bb->error());
    BbdhState st;
    if (h->bb->state(&st, h->stream)) return fail(DVBS2_EDEVICE, h->bb->error());
    if (st.produced > 0) {
        HCHK(hipMemcpyAsync(ts_out, h->d_out, (size_t)st.produced, hipMemcpyDeviceToHost, h->stream));
        HCHK(hipStreamSynchronize(h->stream));
    }
    if (produced) *produced = (int64_t)st.produced;
    return DVBS2_OK;
    API_CATCH
}

} // extern "C"
