/*
 * dvbs2_fec_hip.h -- C ABI of libdvbs2_fec_hip.so: the MI355X (gfx950) DVB-S2/S2X FEC decode path
 * (soft demapper -> layered LDPC -> BCH) as a drop-in for the compute inside the reference's three
 * GNU Radio blocks. Plain pointers and sizes only; no C++ exceptions cross this boundary; every
 * object is a handle (re-entrant across block instances, unlike the reference's global per-TU
 * LdpcDecoder, lib/ldpc_decoder/ldpc_decoder_avx2.cc:21). One caller thread per handle at a time
 * (GNU Radio calls a block's general_work from exactly one thread).
 *
 * Enumerations take the integer values of the reference's include/gnuradio/dvbs2rx/dvb_config.h
 * (dvb_standard_t :15-18, dvb_code_rate_t :20-72, dvb_framesize_t :74-78, dvb_constellation_t :80-101,
 * dvb_outputmode_t :113-116), so the blocks can pass their constructor arguments through unchanged.
 *
 * All functions return DVBS2_OK (0) or a negative DVBS2_E* code; dvbs2_last_error() gives the text.
 * "_device" variants take DEVICE pointers (HBM-resident buffers) and a hipStream_t passed as void*;
 * the plain variants take HOST pointers and stage through buffers owned by the handle.
 * There is no CPU fallback: without a usable HIP device the create calls fail with DVBS2_EDEVICE.
 */
#ifndef DVBS2_FEC_HIP_H
#define DVBS2_FEC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVBS2_OK 0
#define DVBS2_EINVAL (-1)   /* bad argument / unsupported (standard, framesize, rate) */
#define DVBS2_EDEVICE (-2)  /* HIP error or no device */
#define DVBS2_ESIZE (-3)    /* n_frames exceeds the handle's max_frames */

/* dvb_standard_t */
#define DVBS2_STANDARD_DVBS2 0
#define DVBS2_STANDARD_DVBT2 1
/* dvb_framesize_t */
#define DVBS2_FECFRAME_SHORT 0
#define DVBS2_FECFRAME_NORMAL 1
#define DVBS2_FECFRAME_MEDIUM 2
/* dvb_outputmode_t */
#define DVBS2_OM_CODEWORD 0
#define DVBS2_OM_MESSAGE 1
/* dvb_constellation_t (dvb_config.h:80-101: QPSK 0, 16QAM 1, 64QAM 2, 256QAM 3, 8PSK 4, ...); only the two the
 * reference demapper supports are accepted (lib/xfecframe_demapper_cb_impl.cc:45-72). Pinned against the reference
 * header by tests/test_oracle_kat.py::test_enums_match_reference (tests/golden/dvb_config_enums.json). */
#define DVBS2_MOD_QPSK 0
#define DVBS2_MOD_8PSK 4

const char* dvbs2_last_error(void);
int dvbs2_device_count(void);

/* Page-lock a host buffer that the block hands to the plain (host-pointer) entry points again and again -- e.g. a GNU
 * Radio input buffer, once, at start() -- so that the transfers run at PCIe speed and asynchronously (pageable memory
 * goes through a staging copy at a fraction of it and blocks the calling thread). Optional; undo before freeing. Register whole
 * mappings of their own (an mmap'ed buffer), not pieces of the malloc heap; where the caller can choose its memory: dvbs2_host_alloc below.
 * dvbs2_ldpc_decode() also lands its results directly in bits_out / llr_out / ret when THOSE are page-locked (registered here or
 * allocated with hipHostMalloc) instead of in pinned buffers of the handle that it copies out afterwards -- the WHOLE output range has
 * to lie inside ONE registration / allocation visible to the handle's device (a range that spans two registrations with a pageable
 * hole, or any range the runtime cannot vouch for, goes through the handle's pinned buffers: slower, never wrong).
 * Device pointers handed to the *_device entry points: d_llr_in and d_llr_out are accessed with 8-byte loads / stores (align them
 * to 8 bytes: hipMalloc'ed buffers and whole-frame offsets into them are, N is a multiple of 8); a misaligned one is refused with
 * DVBS2_EINVAL. */
int dvbs2_host_register(void* p, size_t bytes);
int dvbs2_host_unregister(void* p);
/* Page-locked host memory allocated BY THE DRIVER (hipHostMalloc / hipHostFree) for callers that do not link HIP themselves: what a
 * GNU Radio >= 3.10 custom buffer allocator or a block's own staging buffer should use where it can choose. Preferred
 * over dvbs2_host_register on ordinary malloc'ed memory: a registration mirrors pages the kernel's memory management still owns
 * (transparent huge pages, compaction, fork), and round 6 saw a GPU write into registered heap memory of a long-running process fault
 * ("write access to a read-only page", Linux 6.18 with transparent_hugepage=always, intermittent); driver-allocated memory is not
 * subject to that. *p is 4096-byte aligned. (reference side: the item buffers of lib/ldpc_decoder_bb_impl.cc:394-455 / bch_decoder_bb_impl.cc:84-117) */
int dvbs2_host_alloc(void** p, size_t bytes);
int dvbs2_host_free(void* p);
/* 1 when [p, p + bytes) lies inside ONE page-locked allocation / registration as the runtime records it (the test the host-buffer
 * entry applies to the caller's buffers before it lets the copy engine address them directly), else 0. Diagnostics and tests. */
int dvbs2_host_is_page_locked(const void* p, size_t bytes);

/* ---- parameter map: replaces get_fec_info(), reference lib/fec_params.h:36-39 / fec_params.cc:16-344,
 * plus the table selection of lib/ldpc_decoder_bb_impl.cc:104-307 ---- */
typedef struct {
    uint32_t bch_k, bch_n, bch_t; /* fec_info_t::bch */
    uint32_t ldpc_k, ldpc_n;      /* fec_info_t::ldpc (ldpc_k == bch_n) */
    uint32_t table_k;             /* K of the LDPC parity table actually used */
    char table[24];               /* e.g. "S2_TABLE_B4" */
} dvbs2_fec_info_t;
int dvbs2_get_fec_info(int standard, int framesize, int rate, dvbs2_fec_info_t* out);
/* rate enumerator name ("C1_2", ...) or NULL; dvbs2_rate_from_name returns -1 when unknown */
const char* dvbs2_rate_name(int rate);
int dvbs2_rate_from_name(const char* name);

/* ---- LDPC schedule introspection (host only, no device needed): the (group, shift) entries of one
 * layer as derived from the accumulator-address table; used by the tests of the schedule compiler.
 * Returns the number of data entries of the layer (<0 on error). groups/shifts may be NULL. ---- */
int dvbs2_ldpc_table_info(const char* table, int* n, int* k, int* q, int* links_total, int* conflict_layers);
/* names of the built-in tables ("S2_TABLE_B4", "S2X_TABLE_C8", "T2_TABLE_A3", ...: the reference's DVB_*_TABLE_*
 * structs, lib/dvb_s2_tables.hh, dvb_s2x_tables.hh, dvb_t2_tables.hh) for index 0, 1, ...; NULL past the end */
const char* dvbs2_ldpc_table_name(int index);
int dvbs2_ldpc_layer_info(const char* table, int layer, int* block, int* groups, int* shifts, int max_entries);

/* ---- LDPC: replaces ldpc_*::ldpc_dec_init + ldpc_*::ldpc_dec_decode as called by
 * ldpc_decoder_bb_impl (reference lib/ldpc_decoder_bb_impl.cc:34-52, :320-347, :406-442) ----
 * group_size G = frames that share one iteration count = the reference's d_simd_size (32 with AVX2,
 * 16 otherwise, :312-345); frames [G*g, G*g+G) form group g. G = 1 decodes every frame on its own. */
typedef struct dvbs2_ldpc dvbs2_ldpc_t;
int dvbs2_ldpc_create(dvbs2_ldpc_t** h, int standard, int framesize, int rate,
                      int group_size, int max_frames, int device);
/* same, selecting a parity table by name; message_bits = bits emitted in DVBS2_OM_MESSAGE mode */
int dvbs2_ldpc_create_table(dvbs2_ldpc_t** h, const char* table, int message_bits,
                            int group_size, int max_frames, int device);
void dvbs2_ldpc_destroy(dvbs2_ldpc_t* h);
int dvbs2_ldpc_params(const dvbs2_ldpc_t* h, int* n, int* table_k, int* message_bits, int* q, int* group_size);
/*
 * llr_in       n_frames * N int8, frame-major, positive = bit 0 (the block's input stream, :407)
 * max_trials   iteration cap (> 0; the block maps 0 to 25 before calling, :391,402)
 * out_mode     DVBS2_OM_MESSAGE -> message_bits/8 bytes per frame, DVBS2_OM_CODEWORD -> N/8 (:404)
 * bits_out     hard decisions, MSB first (:432-442)
 * llr_out      NULL or n_frames * N decoded LLRs (the llr_pdu payload, :422-429)
 * ret          NULL or one int32 per group: what decode() returned for that batch -- trials left
 *              (max_trials - updates), or -1 when the cap was hit without convergence (:410-419)
 * n_frames need not be a multiple of G; a trailing partial group is a group of its own.
 * The batch-coupled stopping rule (every frame of a group runs exactly as many updates as the reference's SIMD batch, :153 of
 * lib/ldpc_decoder/layered_decoder.hh) is resolved on the device, inside the first launch for groups of up to 64 frames (the frames of a
 * group agree after every syndrome test); results do not depend on how frames are scheduled.
 */
int dvbs2_ldpc_decode(dvbs2_ldpc_t* h, const int8_t* llr_in, int n_frames, int max_trials, int out_mode,
                      uint8_t* bits_out, int8_t* llr_out, int32_t* ret);
int dvbs2_ldpc_decode_device(dvbs2_ldpc_t* h, const int8_t* d_llr_in, int n_frames, int max_trials,
                             int out_mode, uint8_t* d_bits_out, int8_t* d_llr_out, int32_t* d_ret,
                             void* stream);
/* The same decode WITHOUT host synchronisation: everything (first pass, the device-side resolution of the batch-coupled
 * stopping rule, the output stage) is enqueued on `stream` and the call returns; dvbs2_ldpc_finish() waits for the stream
 * and completes the rare group that needed more rounds than were enqueued. Outputs are final once finish() returned
 * DVBS2_OK; one decode may be outstanding per handle. dvbs2_ldpc_decode_device == enqueue + finish. This is what lets a
 * block overlap the transfers and neighbours of batch k + 1 with the LDPC of batch k
 * (reference call site: lib/ldpc_decoder_bb_impl.cc:406-449, one blocking call per SIMD batch).
 * Two handles (own state each) on two streams, one call in flight on each, also hide the tail of a launch whose batches stop early
 * (measured: 0.89 -> 0.92 of the rate the iteration count allows; INTEGRATION.md "Double buffering"); the handles of one device share
 * what they need to share (the wave-pattern counters of the one-frame kernel builds), nothing else couples them. */
int dvbs2_ldpc_enqueue_device(dvbs2_ldpc_t* h, const int8_t* d_llr_in, int n_frames, int max_trials,
                              int out_mode, uint8_t* d_bits_out, int8_t* d_llr_out, int32_t* d_ret,
                              void* stream);
int dvbs2_ldpc_finish(dvbs2_ldpc_t* h);
/* HIP-event timing of the dominant kernel (the layered update sweep) on its launch stream.
 * enable != 0 starts/reset accumulation; reads back total milliseconds and launch count. */
int dvbs2_ldpc_profile(dvbs2_ldpc_t* h, int enable, double* total_ms, int* launches);
/* Diagnostics. Host-driven resolution rounds since the handle was created: the group-synchronous stopping rule is resolved inside
 * the first launch; a frame that waited longer than the give-up threshold for its group makes finish() complete the group with resume
 * launches (results identical). Zero in normal operation -- tests and bench.py assert it. */
int dvbs2_ldpc_fallback_rounds(const dvbs2_ldpc_t* h);
/* Diagnostics. Plain hipMemcpyAsync rate of this box's host link, GB/s, best of two timed repetitions: `bytes` split evenly over
 * n_streams (1..16) concurrent streams, host memory kind 0 = hipHostMalloc, 1 = a mapping of its own + hipHostRegister (what dvbs2_host_register
 * does to a caller's buffer), 2 = pageable malloc. Beside the host-entry rates of bench.py (config2_host): is the link or the
 * pipeline what limits dvbs2_ldpc_decode? (reference call site that hands over host buffers: lib/ldpc_decoder_bb_impl.cc:406-449) */
int dvbs2_measure_host_copy(int device, size_t bytes, int n_streams, int kind, double* h2d_gbs, double* d2h_gbs);
/* Diagnostics. The shader clock of this device under VALU load, GHz: every SIMD runs dependent adds for ~1 ms; the s_memtime delta (what the
 * cycle stamps and the SQ cycle counters count in) over the s_memrealtime delta (100 MHz) of one workgroup. bench.py converts the SQ pass's
 * cycle counts with it instead of assuming the nominal 2.4 GHz (roofline.limiter). kernel_ms (nullable): duration of the probe. */
int dvbs2_measure_shader_clock(int device, double* ghz, double* kernel_ms);
/* Diagnostics / tests. The one-frame sweep kernels keep one array of per-CU wave-pattern counters PER DEVICE, shared by all handles of that
 * device (two workgroups on a CU take complementary patterns whichever handle launched them). Returns the array kept for `device_key`
 * (created on `device` when the key is new) and how many of its words are non-zero (all zero while no sweep kernel runs): the same key gives
 * the same array, another key another one -- exercised with keys a one-GPU box does not have (the multi-GPU split of SURVEY 8(e) runs one
 * process per GPU, but several handles on several devices of ONE process are allowed: INTEGRATION.md). */
int dvbs2_debug_cu_slot_table(int device, int device_key, unsigned long long* table_address, int* nonzero_words);
/* which sweep kernel the handle launches, as rocprofv3 names it: "ldpc_layered_kernel<DMAX>" or
 * "ldpc_layered_pr_kernel" (parity LLRs kept in registers / message records; chosen per table, identical results) */
const char* dvbs2_ldpc_kernel_name(const dvbs2_ldpc_t* h);

/* ---- BCH: replaces bch_codec<uint32_t, bitset256_t>::decode(u8_cptr_t, u8_ptr_t) as called by
 * bch_decoder_bb_impl (reference lib/bch.h:151, lib/bch_decoder_bb_impl.cc:58-66, :94-113) ----
 * The field is chosen like the block does: GF(2^16) x^16+x^5+x^3+x^2+1 for normal, GF(2^14) x^14+x^5+x^3+x+1
 * for short, GF(2^15) x^15+x^5+x^3+x^2+1 for medium frames; (n, t) from get_fec_info. */
typedef struct dvbs2_bch dvbs2_bch_t;
int dvbs2_bch_create(dvbs2_bch_t** h, int standard, int framesize, int rate, int max_frames, int device);
/* any binary BCH code over GF(2^m), 3 <= m <= 16, t <= 12, n (0 = 2^m - 1) and k multiples of 8 */
int dvbs2_bch_create_raw(dvbs2_bch_t** h, int m, uint32_t prim_poly, int t, int n, int max_frames, int device);
void dvbs2_bch_destroy(dvbs2_bch_t* h);
int dvbs2_bch_params(const dvbs2_bch_t* h, int* n, int* k, int* t);
/* generator polynomial coefficients (one byte per coefficient, index = power of x), host only;
 * returns deg g or <0. gen may be NULL. */
int dvbs2_bch_genpoly(const dvbs2_bch_t* h, uint8_t* gen, int max_coefs);
/*
 * cw           n_frames * n/8 bytes, network bit order (first bit = x^(n-1), lib/bch.cc:436-449)
 * msg          n_frames * k/8 bytes: the systematic part with the located errors flipped (lib/bch.cc:471,445-450)
 * corrections  per frame: number of corrected bits (>= 0), -1 = more than t errors / roots not all found
 *              (the block counts it in d_frame_error_cnt, lib/bch_decoder_bb_impl.cc:101-107), -2 = the
 *              reference would have thrown here (std::out_of_range from galois_field::get_exponent(0),
 *              lib/gf.h:110 via lib/bch.cc:359-367; or "Error location number out of range", lib/bch.cc:443-444)
 */
int dvbs2_bch_decode(dvbs2_bch_t* h, const uint8_t* cw, int n_frames, uint8_t* msg, int32_t* corrections);
/* Fuse the next block of the flowgraph, bbdescrambler_bb (reference lib/bbdescrambler_bb_impl.cc:67-82,
 * apps/dvbs2-rx:863-864), into the decoder's output stage: msg[j] ^= PRBS[j], j < k/8, per frame. Off by default. */
int dvbs2_bch_set_descramble(dvbs2_bch_t* h, int enable);
/* the BBFRAME energy-dispersal sequence itself (1 + x^14 + x^15, register 100101010000000, packed MSB first;
 * reference init_bb_derandomiser(), lib/bbdescrambler_bb_impl.cc:51-65), host only; n_bytes <= 8100 */
int dvbs2_bb_descramble_sequence(uint8_t* seq, int n_bytes);
/* Device pointers, asynchronous on `stream`. The handle owns the batch's syndrome words (batches of 32 frames and more compute the odd
 * syndromes of all frames as one GF(2) matrix product before the per-frame stage): calls of ONE handle must be ordered -- the same
 * stream, or synchronised by the caller --; concurrent batches take one handle each. */
int dvbs2_bch_decode_device(dvbs2_bch_t* h, const uint8_t* d_cw, int n_frames, uint8_t* d_msg,
                            int32_t* d_corrections, void* stream);

/* ---- soft demapper: replaces QpskConstellation::demap_soft (reference lib/qpsk.h:208-214) and the
 * PhaseShiftKeying<8>::soft loop + column de-interleave (lib/psk.hh:143-150,
 * lib/xfecframe_demapper_cb_impl.cc:152-176) inside xfecframe_demapper_cb_impl::general_work ----
 * constellation: DVBS2_MOD_QPSK or DVBS2_MOD_8PSK; anything else fails with DVBS2_EINVAL
 * ("Unsupported constellation", lib/xfecframe_demapper_cb_impl.cc:70-72). */
typedef struct dvbs2_demap dvbs2_demap_t;
int dvbs2_demap_create(dvbs2_demap_t** h, int framesize, int rate, int constellation, int max_frames, int device);
void dvbs2_demap_destroy(dvbs2_demap_t* h);
/* symbols per frame (d_xfecframe_len), LLRs per frame (d_fecframe_len), bits per symbol, 8PSK column
 * order (0 = "012", 1 = "210", 2 = "102") */
int dvbs2_demap_params(const dvbs2_demap_t* h, int* n_syms, int* n_llr, int* n_mod, int* column_order);
/*
 * syms      n_frames * n_syms complex symbols as interleaved (re, im) floats (gr_complex layout)
 * n0        noise energy N0 = Es/SNR (the block's d_N0, lib/xfecframe_demapper_cb_impl.cc:146-148,313-315):
 *           n0_count == 1: one value for all frames; n0_count == n_frames: one per frame
 * llr_out   n_frames * n_llr int8 LLRs, natural bit order (8PSK: de-interleaved)
 */
int dvbs2_demap_soft(dvbs2_demap_t* h, const float* syms, int n_frames, const float* n0, int n0_count,
                     int8_t* llr_out);
int dvbs2_demap_soft_device(dvbs2_demap_t* h, const float* d_syms, int n_frames, const float* d_n0,
                            int n0_count, int8_t* d_llr_out, void* stream);
/* pre-decoder linear SNR estimate per frame (lib/xfecframe_demapper_cb_impl.cc:128-149, lib/qpsk.h:240-244).
 * Float reduction in a different order than the reference: equal within tolerance, not bit-exact. */
int dvbs2_demap_estimate_snr(dvbs2_demap_t* h, const float* syms, int n_frames, float* snr_lin);
int dvbs2_demap_estimate_snr_device(dvbs2_demap_t* h, const float* d_syms, int n_frames, float* d_snr_lin,
                                    void* stream);
/* post-decoder refinement, the per-frame body of handle_llr_pdu() (lib/xfecframe_demapper_cb_impl.cc:268-307;
 * QPSK: QpskConstellation::estimate_snr(in, ref_llrs, n) lib/qpsk.h:266-281): the reference constellation point of
 * every symbol is re-mapped from the signs of the decoded LLRs (ref_llr: n_frames * n_llr int8 in the decoder's
 * natural bit order, LLR < 0 => bit 1; 8PSK bits are taken through the column interleaver), then
 * snr = sum|ref|^2 / sum|x - ref|^2 per frame. The block averages the per-frame values of one llr_pdu and sets
 * N0 = 1 / mean (:309-315). Same tolerance note as above. */
int dvbs2_demap_refine_snr(dvbs2_demap_t* h, const float* syms, const int8_t* ref_llr, int n_frames,
                           float* snr_lin);
int dvbs2_demap_refine_snr_device(dvbs2_demap_t* h, const float* d_syms, const int8_t* d_ref_llr, int n_frames,
                                  float* d_snr_lin, void* stream);

/* ---- whole chain on the device: xfecframe_demapper_cb -> ldpc_decoder_bb (OM_MESSAGE) -> bch_decoder_bb,
 * as wired in apps/dvbs2-rx:853-863; intermediate LLRs and LDPC output stay in HBM ---- */
typedef struct dvbs2_chain dvbs2_chain_t;
int dvbs2_chain_create(dvbs2_chain_t** h, int standard, int framesize, int rate, int constellation,
                       int group_size, int max_frames, int device);
void dvbs2_chain_destroy(dvbs2_chain_t* h);
/* bytes per frame out (bch k / 8), symbols per frame in */
int dvbs2_chain_params(const dvbs2_chain_t* h, int* n_syms, int* msg_bytes);
/* The chain from LLRs (ldpc_decoder_bb -> bch_decoder_bb, apps/dvbs2-rx:857-863): for constellations whose soft demapper
 * is not part of the reference (lib/xfecframe_demapper_cb_impl.cc:70-72 rejects everything but QPSK and 8PSK) the LLRs
 * come from elsewhere; this is BASELINE config "9/10 normal" run from int8 LLRs. */
int dvbs2_chain_create_llr(dvbs2_chain_t** h, int standard, int framesize, int rate, int group_size, int max_frames,
                           int device);
/* LLRs per frame in (N), bytes per frame out (bch k / 8), group size */
int dvbs2_chain_llr_params(const dvbs2_chain_t* h, int* n_llr, int* msg_bytes, int* group_size);
/* also apply bbdescrambler_bb in the BCH output stage (dvbs2_bch_set_descramble) */
int dvbs2_chain_set_descramble(dvbs2_chain_t* h, int enable);
/* d_msg: n_frames * bch_k/8; d_ldpc_ret (nullable): one per LDPC group; d_bch_corr (nullable -> internal): per frame */
int dvbs2_chain_decode_device(dvbs2_chain_t* h, const float* d_syms, int n_frames, const float* d_n0, int n0_count,
                              int max_trials, uint8_t* d_msg, int32_t* d_ldpc_ret, int32_t* d_bch_corr, void* stream);

/* d_llr: n_frames * N int8 LLRs (works on chains of either kind) */
int dvbs2_chain_decode_llr_device(dvbs2_chain_t* h, const int8_t* d_llr, int n_frames, int max_trials, uint8_t* d_msg,
                                  int32_t* d_ldpc_ret, int32_t* d_bch_corr, void* stream);
/* enqueue-only variants + finish, as for the LDPC decoder: all kernels of the call go to `stream` (demapper, LDPC,
 * BCH back to back, nothing waits on the host), one call may be outstanding per handle */
int dvbs2_chain_enqueue_device(dvbs2_chain_t* h, const float* d_syms, int n_frames, const float* d_n0, int n0_count,
                               int max_trials, uint8_t* d_msg, int32_t* d_ldpc_ret, int32_t* d_bch_corr, void* stream);
int dvbs2_chain_enqueue_llr_device(dvbs2_chain_t* h, const int8_t* d_llr, int n_frames, int max_trials, uint8_t* d_msg,
                                   int32_t* d_ldpc_ret, int32_t* d_bch_corr, void* stream);
int dvbs2_chain_finish(dvbs2_chain_t* h);
/* The fused chain from HOST buffers (the "fused entry, symbols -> message bytes" of SURVEY 8(b)): what the three blocks do with the item
 * buffers GNU Radio hands them, in one call -- xfecframe_demapper_cb_impl::general_work (reference lib/xfecframe_demapper_cb_impl.cc:101-186)
 * -> ldpc_decoder_bb_impl::general_work (lib/ldpc_decoder_bb_impl.cc:394-455) -> bch_decoder_bb_impl::general_work
 * (lib/bch_decoder_bb_impl.cc:84-117), wired as in apps/dvbs2-rx:853-863.
 * syms      n_frames * n_syms complex symbols as interleaved (re, im) floats (HOST), n0 / n0_count as for dvbs2_demap_soft (HOST)
 * msg       n_frames * bch_k/8 bytes (HOST); ldpc_ret (nullable): one int32 per LDPC group; bch_corr (nullable): one int32 per frame
 * The call is cut into chunks of whole LDPC groups over four streams (the plan of dvbs2_ldpc_decode): the input copy of chunk c + 1 and
 * the output copy of chunk c - 1 run under the kernels of chunk c. Buffers that lie inside ONE page-locked allocation / registration
 * (dvbs2_host_register, hipHostMalloc) are addressed by the copy engine directly; pageable ones go through staging (slower, never wrong).
 * An 8PSK normal frame is 172.8 KB of symbols: the host link (~57 GB/s measured) bounds this entry near 320 k frames/s. */
int dvbs2_chain_decode(dvbs2_chain_t* h, const float* syms, int n_frames, const float* n0, int n0_count, int max_trials,
                       uint8_t* msg, int32_t* ldpc_ret, int32_t* bch_corr);
/* the same from int8 LLRs on the host (chains of either kind): ldpc_decoder_bb -> bch_decoder_bb */
int dvbs2_chain_decode_llr(dvbs2_chain_t* h, const int8_t* llr, int n_frames, int max_trials, uint8_t* msg, int32_t* ldpc_ret,
                           int32_t* bch_corr);
/* dvbs2_ldpc_profile / dvbs2_ldpc_kernel_name of the chain's LDPC stage (the dominant kernel) */
int dvbs2_chain_ldpc_profile(dvbs2_chain_t* h, int enable, double* total_ms, int* launches);
const char* dvbs2_chain_ldpc_kernel_name(const dvbs2_chain_t* h);

/* ---- upstream neighbour: the PLFRAME payload step of plsync_cc_impl::handle_payload()
 * (reference lib/plsync_cc_impl.cc:644-653, :727-795): PL descrambling (lib/pl_descrambler.cc:36-105), pilot
 * block removal (:480-485) and phase de-rotation, restarted at every 16-slot segment of a coarse-corrected frame
 * from the preceding pilot block's phase estimate (:759-763). What stays in the block: frame/frequency
 * synchronisation, i.e. everything that PRODUCES the per-frame parameters below.
 * gold_code   PL scrambling code n (0 .. 2^18-2); n_slots 36..360 (lib/pl_signaling.cc:28-48)
 * payload     n_frames * payload_len complex symbols (re, im), payload_len = 90 n_slots + 36 n_pilots,
 *             n_pilots = has_pilots ? (n_slots - 1) / 16 : 0 (lib/pl_signaling.cc:51-60)
 * plheader_phase[f], phase_inc[f] = 2 pi fine_foffset (used only when coarse_corrected[f] != 0, :730-732),
 * pilot_phase[f * n_pilots + i] = pl_freq_sync::get_pilot_phase(i)
 * xfecframes  n_frames * 90 n_slots complex symbols: the input of dvbs2_demap_soft
 * The rotator is a float recurrence in the reference (VOLK); here the phase of every symbol is evaluated directly:
 * equal within 1e-4 absolute per component for unit-energy symbols, not bit-exact. */
typedef struct dvbs2_plpayload dvbs2_plpayload_t;
int dvbs2_plpayload_create(dvbs2_plpayload_t** h, int gold_code, int n_slots, int has_pilots, int max_frames, int device);
void dvbs2_plpayload_destroy(dvbs2_plpayload_t* h);
int dvbs2_plpayload_params(const dvbs2_plpayload_t* h, int* payload_len, int* xfecframe_len, int* n_pilots);
int dvbs2_plpayload_process(dvbs2_plpayload_t* h, const float* payload, int n_frames, const float* plheader_phase,
                            const float* phase_inc, const int32_t* coarse_corrected, const float* pilot_phase,
                            float* xfecframes);
int dvbs2_plpayload_process_device(dvbs2_plpayload_t* h, const float* d_payload, int n_frames,
                                   const float* d_plheader_phase, const float* d_phase_inc,
                                   const int32_t* d_coarse_corrected, const float* d_pilot_phase, float* d_xfecframes,
                                   void* stream);
/* the PL scrambling sequence Rn(i) in 0..3 (ETSI EN 302 307-1 clause 5.5.4; reference lib/pl_descrambler.cc:62-98),
 * host only; n <= 33192 */
int dvbs2_pl_scrambling_rn(int gold_code, uint8_t* rn, int n);

/* ---- downstream neighbour: BBFRAME de-header, replaces bbdeheader_bb_impl::general_work (reference
 * lib/bbdeheader_bb_impl.cc:144-264) with parse_bbheader (:77-136) and check_crc8 (:138-142, generator
 * x^8 + x^7 + x^6 + x^4 + x^2 + 1, :55). Input: whole descrambled BBFRAMEs of kbch / 8 bytes (what dvbs2_bch_decode /
 * dvbs2_chain_* emit with descrambling on); output: 188-byte MPEG-TS packets back to back, sync byte restored, transport
 * error indicator set where the packet's CRC-8 fails. The block's state -- synchronised flag, the partial TS packet that
 * continues in the next BBFRAME, the five counters -- lives in the handle and carries over from call to call exactly as it
 * carries over between work() calls of the block; dvbs2_bbdeheader_reset() gives the block as constructed.
 * One defined deviation: a header that passes every check but has SYNCD/8 + 1 > DFL/8 while the block re-synchronises makes
 * the reference's unsigned byte count wrap and its loop read past the buffer (:201-202); such a BBFRAME is dropped here and
 * counted in `overruns`.
 * kbch_bits = fec_info.bch_k of (standard, framesize, rate); out capacity >= n_frames * max_out_bytes_per_frame. */
typedef struct dvbs2_bbdeheader dvbs2_bbdeheader_t;
typedef struct {
    uint64_t packets, errors, bbframes, dropped, gaps; /* d_packet_cnt, d_error_cnt, d_bbframe_cnt, d_bbframe_drop_cnt, d_bbframe_gap_cnt */
    uint64_t overruns;
    int32_t synched, partial_ts_bytes;                  /* d_synched, d_partial_ts_bytes */
} dvbs2_bbdeheader_counters_t;
int dvbs2_bbdeheader_create(dvbs2_bbdeheader_t** h, int standard, int framesize, int rate, int max_frames, int device);
int dvbs2_bbdeheader_create_raw(dvbs2_bbdeheader_t** h, int kbch_bits, int max_frames, int device);
void dvbs2_bbdeheader_destroy(dvbs2_bbdeheader_t* h);
int dvbs2_bbdeheader_params(const dvbs2_bbdeheader_t* h, int* kbch_bytes, int* max_dfl_bits, int* max_out_bytes_per_frame);
/* host buffers; *produced = bytes written to ts_out (a multiple of 188) */
int dvbs2_bbdeheader_process(dvbs2_bbdeheader_t* h, const uint8_t* bbframes, int n_frames, uint8_t* ts_out, int64_t* produced);
/* device buffers, asynchronous on `stream`; the byte count of the call is read with dvbs2_bbdeheader_finish */
int dvbs2_bbdeheader_process_device(dvbs2_bbdeheader_t* h, const uint8_t* d_bbframes, int n_frames, uint8_t* d_ts_out, void* stream);
/* waits for the calls enqueued on `stream`; *produced = bytes written by the LAST of them */
int dvbs2_bbdeheader_finish(dvbs2_bbdeheader_t* h, int64_t* produced, void* stream);
int dvbs2_bbdeheader_counters(dvbs2_bbdeheader_t* h, dvbs2_bbdeheader_counters_t* out, void* stream);
int dvbs2_bbdeheader_reset(dvbs2_bbdeheader_t* h, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DVBS2_FEC_HIP_H */
