// dvbs2rx_hip_blocks.h -- host-side mirror (C++17, header only) of the reference blocks on the FEC hot
// path, implemented over the C ABI of libdvbs2_fec_hip.so. Same class names, make() argument order and
// meaning, forecast()/general_work() item accounting, getters and error behaviour as the reference:
//
//   gr::dvbs2rx::ldpc_decoder_bb        include/gnuradio/dvbs2rx/ldpc_decoder_bb.h:37-50, lib/ldpc_decoder_bb_impl.cc
//   gr::dvbs2rx::bch_decoder_bb         include/gnuradio/dvbs2rx/bch_decoder_bb.h,        lib/bch_decoder_bb_impl.cc
//   gr::dvbs2rx::xfecframe_demapper_cb  include/gnuradio/dvbs2rx/xfecframe_demapper_cb.h, lib/xfecframe_demapper_cb_impl.cc
//   gr::dvbs2rx::bbdeheader_bb          include/gnuradio/dvbs2rx/bbdeheader_bb.h,         lib/bbdeheader_bb_impl.cc
//
// It deliberately does NOT depend on GNU Radio (absent from the build image): the classes expose the
// gr::block work-function signature with plain std::vector arguments, so that the reference's *_impl classes can
// either derive from these or forward to them (INTEGRATION.md shows the two-line patch). What GNU Radio itself
// provides (scheduler, buffers, message ports) stays in the reference; the llr_pdu hand-off is a callback here.
#pragma once
#include <cstdint>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/dvbs2_fec_hip.h"

namespace dvbs2rx_hip {

// enumerations with the reference's values (dvb_config.h)
enum dvb_standard_t { STANDARD_DVBS2 = 0, STANDARD_DVBT2 };
enum dvb_framesize_t { FECFRAME_SHORT = 0, FECFRAME_NORMAL, FECFRAME_MEDIUM };
enum dvb_constellation_t { MOD_QPSK = 0, MOD_16QAM, MOD_64QAM, MOD_256QAM, MOD_8PSK, MOD_8APSK, MOD_16APSK, MOD_8_8APSK, MOD_32APSK };
enum dvb_outputmode_t { OM_CODEWORD = 0, OM_MESSAGE };
enum dvb_infomode_t { INFO_OFF = 0, INFO_ON };
typedef int dvb_code_rate_t; // use dvbs2_rate_from_name("C1_2") or the reference's enumerator value

typedef std::vector<int> gr_vector_int;
typedef std::vector<const void*> gr_vector_const_void_star;
typedef std::vector<void*> gr_vector_void_star;

inline void check(int rc)
{
    if (rc != DVBS2_OK) throw std::runtime_error(dvbs2_last_error());
}

// ---------------------------------------------------------------------------------------------- LDPC
class ldpc_decoder_bb {
public:
    typedef std::shared_ptr<ldpc_decoder_bb> sptr;
    // batch_frames replaces the reference's d_simd_size as the scheduling granule: general_work consumes whole
    // multiples of it; group_size keeps the reference's batch-coupled stopping rule (32 = AVX2, 16 otherwise).
    static sptr make(dvb_standard_t standard, dvb_framesize_t framesize, dvb_code_rate_t rate,
                     dvb_constellation_t /*constellation*/, dvb_outputmode_t outputmode, dvb_infomode_t /*infomode*/,
                     int max_trials, int /*debug_level*/ = 0, int group_size = 32, int batch_frames = 512, int device = 0)
    {
        return sptr(new ldpc_decoder_bb(standard, framesize, rate, outputmode, max_trials, group_size, batch_frames, device));
    }
    ~ldpc_decoder_bb() { dvbs2_ldpc_destroy(d_h); }

    // lib/ldpc_decoder_bb_impl.cc:380-389
    void forecast(int noutput_items, gr_vector_int& ninput_items_required) const
    {
        if (d_output_mode == OM_MESSAGE) ninput_items_required[0] = (noutput_items / (int)d_kldpc_bytes) * (int)d_nldpc;
        else ninput_items_required[0] = 8 * noutput_items;
    }
    int output_multiple() const { return (int)((d_output_mode ? d_kldpc_bytes : d_nldpc_bytes) * (unsigned)d_group); }

    // lib/ldpc_decoder_bb_impl.cc:394-455. Returns the items produced; *consumed gets consume_each()'s argument.
    int general_work(int noutput_items, gr_vector_int& /*ninput_items*/, gr_vector_const_void_star& input_items,
                     gr_vector_void_star& output_items, int* consumed)
    {
        const int8_t* in = static_cast<const int8_t*>(input_items[0]);
        unsigned char* out = static_cast<unsigned char*>(output_items[0]);
        const int trials = d_max_trials == 0 ? 25 : d_max_trials; // DEFAULT_TRIALS, :391,402
        const int output_size = (int)(d_output_mode ? d_kldpc_bytes : d_nldpc_bytes);
        int n_frames = noutput_items / output_size;
        n_frames -= n_frames % d_group;
        int done = 0;
        while (done < n_frames) {
            const int nf = std::min(n_frames - done, d_batch);
            const int ng = nf / d_group;
            d_ret.resize(ng);
            int8_t* llr_out = nullptr;
            if (d_llr_pdu) { d_soft.resize((size_t)nf * d_nldpc); llr_out = d_soft.data(); }
            check(dvbs2_ldpc_decode(d_h, in + (size_t)done * d_nldpc, nf, trials, d_output_mode, out + (size_t)done * output_size,
                                    llr_out, d_ret.data()));
            for (int g = 0; g < ng; g++) { // :411-419
                d_total_trials += d_ret[g] < 0 ? trials : trials - d_ret[g];
                if (d_llr_pdu) d_llr_pdu(d_frame_cnt, d_group, llr_out + (size_t)g * d_group * d_nldpc, (size_t)d_group * d_nldpc); // :422-429
                d_frame_cnt += d_group;
                d_batch_cnt++;
            }
            done += nf;
        }
        *consumed = n_frames * (int)d_nldpc;
        return n_frames * output_size;
    }
    unsigned int get_average_trials() const { return d_batch_cnt ? (unsigned)(d_total_trials / d_batch_cnt) : 0; } // .h:63
    // the llr_pdu message port: (frame_cnt, simd_size, decoded LLRs, count)
    void set_llr_pdu_handler(std::function<void(uint64_t, int, const int8_t*, size_t)> f) { d_llr_pdu = std::move(f); }

private:
    ldpc_decoder_bb(dvb_standard_t standard, dvb_framesize_t framesize, dvb_code_rate_t rate, dvb_outputmode_t outputmode,
                    int max_trials, int group_size, int batch_frames, int device)
        : d_output_mode(outputmode), d_max_trials(max_trials), d_group(group_size), d_batch(batch_frames - batch_frames % group_size)
    {
        if (d_batch < d_group) d_batch = d_group;
        dvbs2_fec_info_t fi;
        check(dvbs2_get_fec_info(standard, framesize, rate, &fi));
        d_kldpc = fi.ldpc_k; d_nldpc = fi.ldpc_n; d_kldpc_bytes = d_kldpc / 8; d_nldpc_bytes = d_nldpc / 8;
        check(dvbs2_ldpc_create(&d_h, standard, framesize, rate, group_size, d_batch, device));
    }
    dvbs2_ldpc_t* d_h = nullptr;
    unsigned d_nldpc = 0, d_nldpc_bytes = 0, d_kldpc = 0, d_kldpc_bytes = 0;
    int d_output_mode, d_max_trials, d_group, d_batch;
    uint64_t d_frame_cnt = 0, d_batch_cnt = 0, d_total_trials = 0;
    std::vector<int32_t> d_ret;
    std::vector<int8_t> d_soft;
    std::function<void(uint64_t, int, const int8_t*, size_t)> d_llr_pdu;
};

// ---------------------------------------------------------------------------------------------- BCH
class bch_decoder_bb {
public:
    typedef std::shared_ptr<bch_decoder_bb> sptr;
    static sptr make(dvb_standard_t standard, dvb_framesize_t framesize, dvb_code_rate_t rate, dvb_outputmode_t /*outputmode*/,
                     int /*debug_level*/ = 0, int batch_frames = 512, int device = 0)
    {
        return sptr(new bch_decoder_bb(standard, framesize, rate, batch_frames, device));
    }
    // also run the next block of the flowgraph, bbdescrambler_bb (lib/bbdescrambler_bb_impl.cc:67-82), in the same kernel
    void set_descramble(bool enable) { check(dvbs2_bch_set_descramble(d_h, enable ? 1 : 0)); }
    ~bch_decoder_bb() { dvbs2_bch_destroy(d_h); }
    void forecast(int noutput_items, gr_vector_int& req) const { req[0] = (noutput_items / d_k_bytes) * d_n_bytes; } // :78-82
    int output_multiple() const { return d_k_bytes; }
    // lib/bch_decoder_bb_impl.cc:84-117. A codeword on which the reference would have thrown (status -2) throws here too.
    int general_work(int noutput_items, gr_vector_int&, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items,
                     int* consumed)
    {
        const unsigned char* in = static_cast<const unsigned char*>(input_items[0]);
        unsigned char* out = static_cast<unsigned char*>(output_items[0]);
        const int n_codewords = noutput_items / d_k_bytes;
        int done = 0;
        while (done < n_codewords) {
            const int nf = std::min(n_codewords - done, d_batch);
            d_corr.resize(nf);
            check(dvbs2_bch_decode(d_h, in + (size_t)done * d_n_bytes, nf, out + (size_t)done * d_k_bytes, d_corr.data()));
            for (int i = 0; i < nf; i++) {
                if (d_corr[i] == -2) throw std::runtime_error("BCH decoder: the reference throws on this codeword (see dvbs2_fec_hip.h)");
                if (d_corr[i] == -1) d_frame_error_cnt++;
                d_frame_cnt++;
            }
            done += nf;
        }
        *consumed = n_codewords * d_n_bytes;
        return n_codewords * d_k_bytes;
    }
    uint64_t get_frame_count() const { return d_frame_cnt; }       // lib/bch_decoder_bb_impl.h:46
    uint64_t get_error_count() const { return d_frame_error_cnt; } // :47

private:
    bch_decoder_bb(dvb_standard_t standard, dvb_framesize_t framesize, dvb_code_rate_t rate, int batch_frames, int device) : d_batch(batch_frames)
    {
        check(dvbs2_bch_create(&d_h, standard, framesize, rate, batch_frames, device));
        int n, k, t;
        check(dvbs2_bch_params(d_h, &n, &k, &t));
        d_n_bytes = n / 8; d_k_bytes = k / 8;
    }
    dvbs2_bch_t* d_h = nullptr;
    int d_n_bytes = 0, d_k_bytes = 0, d_batch;
    uint64_t d_frame_cnt = 0, d_frame_error_cnt = 0;
    std::vector<int32_t> d_corr;
};

// ---------------------------------------------------------------------------------------------- BBFRAME de-header
class bbdeheader_bb {
public:
    typedef std::shared_ptr<bbdeheader_bb> sptr;
    static sptr make(dvb_standard_t standard, dvb_framesize_t framesize, dvb_code_rate_t rate, int /*debug_level*/ = 0,
                     int batch_frames = 512, int device = 0)
    {
        return sptr(new bbdeheader_bb(standard, framesize, rate, batch_frames, device));
    }
    ~bbdeheader_bb() { dvbs2_bbdeheader_destroy(d_h); }
    int output_multiple() const { return d_max_dfl / 8; }                                            // lib/bbdeheader_bb_impl.cc:62
    void forecast(int noutput_items, gr_vector_int& req) const                                       // :70-75
    {
        req[0] = (int)std::ceil((double)noutput_items * 8 / d_max_dfl) * d_kbch_bytes;
    }
    // :144-264. ninput_items[0] bytes are available; whole BBFRAMEs are consumed, whole TS packets produced.
    int general_work(int noutput_items, gr_vector_int& ninput_items, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items,
                     int* consumed)
    {
        const unsigned char* in = static_cast<const unsigned char*>(input_items[0]);
        unsigned char* out = static_cast<unsigned char*>(output_items[0]);
        const int in_bbframes = ninput_items[0] / d_kbch_bytes;
        const int out_bbframes = (int)std::ceil((double)noutput_items * 8 / d_max_dfl);
        const int n_bbframes = std::min(in_bbframes, out_bbframes);                                  // :157-161
        int done = 0, produced = 0;
        while (done < n_bbframes) {
            const int nf = std::min(n_bbframes - done, d_batch);
            d_stage.resize((size_t)nf * d_max_out);
            int64_t n = 0;
            check(dvbs2_bbdeheader_process(d_h, in + (size_t)done * d_kbch_bytes, nf, d_stage.data(), &n));
            std::memcpy(out + produced, d_stage.data(), (size_t)n);
            produced += (int)n; done += nf;
        }
        *consumed = n_bbframes * d_kbch_bytes;
        return produced;
    }
    // lib/bbdeheader_bb_impl.h:89-93
    uint64_t get_packet_count() { return counters().packets; }
    uint64_t get_error_count() { return counters().errors; }
    uint64_t get_bbframe_count() { return counters().bbframes; }
    uint64_t get_bbframe_drop_count() { return counters().dropped; }
    uint64_t get_bbframe_gap_count() { return counters().gaps; }

private:
    bbdeheader_bb(dvb_standard_t standard, dvb_framesize_t framesize, dvb_code_rate_t rate, int batch_frames, int device) : d_batch(batch_frames)
    {
        check(dvbs2_bbdeheader_create(&d_h, standard, framesize, rate, batch_frames, device));
        check(dvbs2_bbdeheader_params(d_h, &d_kbch_bytes, &d_max_dfl, &d_max_out));
    }
    dvbs2_bbdeheader_counters_t counters() { dvbs2_bbdeheader_counters_t c; check(dvbs2_bbdeheader_counters(d_h, &c, nullptr)); return c; }
    dvbs2_bbdeheader_t* d_h = nullptr;
    int d_kbch_bytes = 0, d_max_dfl = 0, d_max_out = 0, d_batch;
    std::vector<unsigned char> d_stage;
};

// ---------------------------------------------------------------------------------------------- demapper
class xfecframe_demapper_cb {
public:
    typedef std::shared_ptr<xfecframe_demapper_cb> sptr;
    static sptr make(dvb_framesize_t framesize, dvb_code_rate_t rate, dvb_constellation_t constellation, int batch_frames = 512, int device = 0)
    {
        return sptr(new xfecframe_demapper_cb(framesize, rate, constellation, batch_frames, device));
    }
    ~xfecframe_demapper_cb() { dvbs2_demap_destroy(d_h); }
    void forecast(int noutput_items, gr_vector_int& req) const { req[0] = noutput_items / d_n_mod; } // :95-99
    int output_multiple() const { return d_fecframe_len; }
    // lib/xfecframe_demapper_cb_impl.cc:101-186. Until the first refined estimate arrives (set_snr / the llr_pdu
    // path of the reference), N0 comes from the pre-decoder estimate of each frame, like d_waiting_first_llr.
    int general_work(int noutput_items, gr_vector_int&, gr_vector_const_void_star& input_items, gr_vector_void_star& output_items, int* consumed)
    {
        const float* in = static_cast<const float*>(input_items[0]); // gr_complex = 2 floats
        int8_t* out = static_cast<int8_t*>(output_items[0]);
        const int n_frames = noutput_items / d_fecframe_len;
        int done = 0;
        while (done < n_frames) {
            const int nf = std::min(n_frames - done, d_batch);
            const float* p = in + (size_t)done * d_xfecframe_len * 2;
            if (d_waiting_first_llr) {
                d_n0.resize(nf);
                check(dvbs2_demap_estimate_snr(d_h, p, nf, d_n0.data()));
                for (int i = 0; i < nf; i++) { d_snr_lin = d_n0[i]; d_n0[i] = 1.0f / d_n0[i]; } // N0 = Es / snr, Es = 1
                check(dvbs2_demap_soft(d_h, p, nf, d_n0.data(), nf, out + (size_t)done * d_fecframe_len));
            } else {
                const float n0 = 1.0f / d_snr_lin;
                check(dvbs2_demap_soft(d_h, p, nf, &n0, 1, out + (size_t)done * d_fecframe_len));
            }
            for (int i = 0; i < nf; i++) { // keep the XFECFRAMEs for the post-decoder refinement (:115-123)
                d_saved[d_pool_idx] = d_frame_cnt + i;
                std::memcpy(&d_pool[d_pool_idx * (size_t)d_xfecframe_len * 2], p + (size_t)i * d_xfecframe_len * 2, (size_t)d_xfecframe_len * 8);
                d_pool_idx = (d_pool_idx + 1) % d_saved.size();
            }
            done += nf; d_frame_cnt += nf;
        }
        *consumed = n_frames * d_xfecframe_len;
        return n_frames * d_fecframe_len;
    }
    float get_snr() const { return 10.0f * std::log10(d_snr_lin); } // lib/xfecframe_demapper_cb_impl.h:74
    // refined (post-decoder) linear SNR, the result of the reference's handle_llr_pdu() (:188-318)
    void set_snr_lin(float snr_lin) { d_snr_lin = snr_lin; d_waiting_first_llr = false; }
    // The llr_pdu message handler (lib/xfecframe_demapper_cb_impl.cc:188-318): frames starting_frame_cnt ..
    // +simd_size-1 are looked up in the pool of saved XFECFRAMEs, the per-frame refined estimates (computed on the
    // GPU in one call) are averaged and become the N0 of the frames demapped from now on. Returns the number of
    // frames that were found (the reference logs and skips the others, :260-266); a malformed PDU is dropped.
    int handle_llr_pdu(uint64_t starting_frame_cnt, int simd_size, const int8_t* llr, size_t n_llr)
    {
        if (n_llr == 0 || simd_size <= 0 || n_llr != (size_t)simd_size * d_fecframe_len) return 0; // :235-239
        std::vector<float> syms; std::vector<int8_t> ref;
        int found = 0;
        for (int i = 0; i < simd_size; i++) {
            size_t idx = d_saved.size();
            for (size_t k = 0; k < d_saved.size(); k++) if (d_saved[k] == starting_frame_cnt + i) { idx = k; break; }
            if (idx == d_saved.size()) continue;
            syms.insert(syms.end(), &d_pool[idx * (size_t)d_xfecframe_len * 2], &d_pool[(idx + 1) * (size_t)d_xfecframe_len * 2]);
            ref.insert(ref.end(), llr + (size_t)i * d_fecframe_len, llr + (size_t)(i + 1) * d_fecframe_len);
            found++;
        }
        float accum = 0;
        std::vector<float> snr(d_batch);
        for (int done = 0; done < found; done += d_batch) {
            const int nf = std::min(found - done, d_batch);
            check(dvbs2_demap_refine_snr(d_h, &syms[(size_t)done * d_xfecframe_len * 2], &ref[(size_t)done * d_fecframe_len], nf, snr.data()));
            for (int i = 0; i < nf; i++) accum += snr[i];
        }
        if (found > 0) { d_snr_lin = accum / found; d_waiting_first_llr = false; } // :309-317
        return found;
    }

private:
    xfecframe_demapper_cb(dvb_framesize_t framesize, dvb_code_rate_t rate, dvb_constellation_t constellation, int batch_frames, int device) : d_batch(batch_frames)
    {
        check(dvbs2_demap_create(&d_h, framesize, rate, constellation, batch_frames, device)); // throws "Unsupported constellation"
        int order;
        check(dvbs2_demap_params(d_h, &d_xfecframe_len, &d_fecframe_len, &d_n_mod, &order));
        d_saved.assign(std::max(64, 2 * batch_frames), std::numeric_limits<uint64_t>::max()); // XFECFRAME_POOL_SIZE, .h:28-32
        d_pool.resize(d_saved.size() * (size_t)d_xfecframe_len * 2);
    }
    dvbs2_demap_t* d_h = nullptr;
    int d_xfecframe_len = 0, d_fecframe_len = 0, d_n_mod = 0, d_batch;
    bool d_waiting_first_llr = true;
    float d_snr_lin = 1.0f;
    uint64_t d_frame_cnt = 0;
    std::vector<float> d_n0;
    std::vector<uint64_t> d_saved; // frame number held by each pool slot
    std::vector<float> d_pool;     // saved XFECFRAMEs, interleaved (re, im)
    size_t d_pool_idx = 0;
};

} // namespace dvbs2rx_hip
