// c_api.hip -- extern "C" boundary (include/dvbs2_fec_hip.h). No exceptions leave this file.
#include "../../include/dvbs2_fec_hip.h"
#include <hip/hip_runtime.h>
#include <cstring>
#include <new>
#include <string>
#include "fec_tables.h"
#include "ldpc_hip.h"
#include "ldpc_schedule.h"

using namespace dvbs2;

static thread_local std::string g_err;
static int fail(int code, const std::string& msg) { g_err = msg; return code; }

#define API_TRY try {
#define API_CATCH } catch (const std::exception& e) { return fail(DVBS2_EDEVICE, e.what()); } catch (...) { return fail(DVBS2_EDEVICE, "unknown exception"); }

struct dvbs2_ldpc {
    LdpcDecoderHip* dec = nullptr;
    // host staging
    int8_t* d_in = nullptr; uint8_t* d_bits = nullptr; int8_t* d_llr = nullptr; int32_t* d_ret = nullptr;
    hipStream_t stream = nullptr;
    int device = 0;
};

extern "C" {

const char* dvbs2_last_error(void) { return g_err.c_str(); }

int dvbs2_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
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
    (void)hipSetDevice(h->device);
    (void)hipFree(h->d_in); (void)hipFree(h->d_bits); (void)hipFree(h->d_llr); (void)hipFree(h->d_ret);
    if (h->stream) (void)hipStreamDestroy(h->stream);
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

int dvbs2_ldpc_decode_device(dvbs2_ldpc_t* h, const int8_t* d_llr_in, int n_frames, int max_trials, int out_mode,
                             uint8_t* d_bits_out, int8_t* d_llr_out, int32_t* d_ret, void* stream)
{
    API_TRY
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    if (n_frames < 0 || max_trials <= 0 || (n_frames && (!d_llr_in || !d_bits_out))) return fail(DVBS2_EINVAL, "bad argument");
    if (out_mode != DVBS2_OM_CODEWORD && out_mode != DVBS2_OM_MESSAGE) return fail(DVBS2_EINVAL, "bad out_mode");
    if (n_frames > h->dec->max_frames()) return fail(DVBS2_ESIZE, "n_frames exceeds max_frames");
    if (h->dec->decode_device(d_llr_in, n_frames, max_trials, out_mode, d_bits_out, d_llr_out, d_ret, (hipStream_t)stream))
        return fail(DVBS2_EDEVICE, h->dec->error());
    return DVBS2_OK;
    API_CATCH
}

int dvbs2_ldpc_decode(dvbs2_ldpc_t* h, const int8_t* llr_in, int n_frames, int max_trials, int out_mode,
                      uint8_t* bits_out, int8_t* llr_out, int32_t* ret)
{
    API_TRY
    if (!h) return fail(DVBS2_EINVAL, "null handle");
    if (n_frames < 0 || max_trials <= 0 || (n_frames && (!llr_in || !bits_out))) return fail(DVBS2_EINVAL, "bad argument");
    if (out_mode != DVBS2_OM_CODEWORD && out_mode != DVBS2_OM_MESSAGE) return fail(DVBS2_EINVAL, "bad out_mode");
    if (n_frames > h->dec->max_frames()) return fail(DVBS2_ESIZE, "n_frames exceeds max_frames");
    if (n_frames == 0) return DVBS2_OK;
#define HCHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) return fail(DVBS2_EDEVICE, std::string(#x) + ": " + hipGetErrorString(e_)); } while (0)
    HCHK(hipSetDevice(h->device));
    const size_t N = h->dec->N(), mf = h->dec->max_frames();
    const int G = h->dec->group_size();
    if (!h->stream) HCHK(hipStreamCreate(&h->stream));
    if (!h->d_in) HCHK(hipMalloc(&h->d_in, mf * N));
    if (!h->d_bits) HCHK(hipMalloc(&h->d_bits, mf * (N / 8)));
    if (!h->d_llr) HCHK(hipMalloc(&h->d_llr, mf * N));
    if (!h->d_ret) HCHK(hipMalloc(&h->d_ret, ((mf + G - 1) / G) * 4));
    const size_t out_bytes = (out_mode ? h->dec->out_bits_message() : (int)N) / 8;
    HCHK(hipMemcpyAsync(h->d_in, llr_in, (size_t)n_frames * N, hipMemcpyHostToDevice, h->stream));
    if (h->dec->decode_device(h->d_in, n_frames, max_trials, out_mode, h->d_bits, llr_out ? h->d_llr : nullptr, h->d_ret, h->stream))
        return fail(DVBS2_EDEVICE, h->dec->error());
    HCHK(hipMemcpyAsync(bits_out, h->d_bits, (size_t)n_frames * out_bytes, hipMemcpyDeviceToHost, h->stream));
    if (llr_out) HCHK(hipMemcpyAsync(llr_out, h->d_llr, (size_t)n_frames * N, hipMemcpyDeviceToHost, h->stream));
    if (ret) HCHK(hipMemcpyAsync(ret, h->d_ret, (size_t)((n_frames + G - 1) / G) * 4, hipMemcpyDeviceToHost, h->stream));
    HCHK(hipStreamSynchronize(h->stream));
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
