"""Python view of the three reference blocks' compute, calling the C ABI.

Names and argument meaning follow the reference block constructors:
  ldpc_decoder_bb.make(standard, framesize, rate, constellation, outputmode, infomode, max_trials, debug_level)
      include/gnuradio/dvbs2rx/ldpc_decoder_bb.h:37-44
The Python classes exist for the tests and the benchmark; the production drop-in is the C++ shim
in INTEGRATION.md that calls the same C entry points from the blocks' general_work().
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import lib, check

__all__ = ["get_fec_info", "rate_id", "LdpcDecoder", "BchDecoder", "Demapper", "FecChain", "BbDeheader", "ldpc_table_info", "ldpc_layer_info",
           "ldpc_table_names", "bb_descramble_sequence", "PlPayload",
           "pl_scrambling_rn", "HostBuffer"]

DEFAULT_TRIALS = 25  # reference lib/ldpc_decoder_bb_impl.cc:391


def rate_id(rate):
    if isinstance(rate, str):
        r = lib.dvbs2_rate_from_name(rate.encode())
        if r < 0:
            raise ValueError(f"unknown code rate {rate}")
        return r
    return int(rate)


def get_fec_info(standard, framesize, rate):
    fi = capi.FecInfo()
    check(lib.dvbs2_get_fec_info(standard, framesize, rate_id(rate), fi))
    return dict(bch_k=fi.bch_k, bch_n=fi.bch_n, bch_t=fi.bch_t, ldpc_k=fi.ldpc_k, ldpc_n=fi.ldpc_n,
                table_k=fi.table_k, table=fi.table.decode())


def ldpc_table_info(table):
    v = [C.c_int() for _ in range(5)]
    check(lib.dvbs2_ldpc_table_info(table.encode(), *v))
    return dict(zip(("N", "K", "q", "links_total", "conflict_layers"), (x.value for x in v)))


def bb_descramble_sequence(n_bytes):
    """The BBFRAME energy-dispersal PRBS as packed bytes (reference lib/bbdescrambler_bb_impl.cc:51-65)."""
    seq = np.zeros(n_bytes, np.uint8)
    check(lib.dvbs2_bb_descramble_sequence(seq.ctypes.data, n_bytes))
    return seq


class HostBuffer:
    """A page-locked host buffer allocated by the driver (dvbs2_host_alloc -> hipHostMalloc) with a numpy view: what a caller of the
    host-pointer entry points should hand over where it can choose its memory (include/dvbs2_fec_hip.h). `array` stays valid until
    free() / the object is dropped."""

    def __init__(self, shape, dtype):
        self.array = None
        self._p = C.c_void_p()
        dt = np.dtype(dtype)
        n = int(np.prod(shape)) * dt.itemsize
        check(lib.dvbs2_host_alloc(C.byref(self._p), max(n, 1)))
        self.nbytes = n
        self.array = np.frombuffer((C.c_char * max(n, 1)).from_address(self._p.value), dtype=dt, count=int(np.prod(shape))).reshape(shape)

    @property
    def ptr(self):
        return self._p.value

    def free(self):
        if self._p:
            self.array = None
            lib.dvbs2_host_free(self._p)
            self._p = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def ldpc_table_names():
    """Names of all built-in LDPC tables (the reference's DVB_*_TABLE_* structs)."""
    out, i = [], 0
    while True:
        n = lib.dvbs2_ldpc_table_name(i)
        if n is None:
            return out
        out.append(n.decode())
        i += 1


def ldpc_layer_info(table, layer):
    g = np.zeros(64, np.int32)
    s = np.zeros(64, np.int32)
    blk = C.c_int()
    cnt = check(lib.dvbs2_ldpc_layer_info(table.encode(), layer, blk, g.ctypes.data, s.ctypes.data, 64))
    return dict(cnt=cnt, block=blk.value, groups=g[:cnt].tolist(), shifts=s[:cnt].tolist())


class LdpcDecoder:
    """ldpc_decoder_bb's compute: batches of int8 LLR frames -> packed hard bits (+ decoded LLRs)."""

    def __init__(self, standard=capi.STANDARD_DVBS2, framesize=capi.FECFRAME_NORMAL, rate="C1_2",
                 outputmode=capi.OM_MESSAGE, max_trials=0, group_size=32, max_frames=64, device=0,
                 table=None, message_bits=None):
        self._h = C.c_void_p()
        if table is not None:
            check(lib.dvbs2_ldpc_create_table(C.byref(self._h), table.encode(), int(message_bits),
                                              group_size, max_frames, device))
        else:
            check(lib.dvbs2_ldpc_create(C.byref(self._h), standard, framesize, rate_id(rate),
                                        group_size, max_frames, device))
        v = [C.c_int() for _ in range(5)]
        check(lib.dvbs2_ldpc_params(self._h, *v))
        self.N, self.K, self.message_bits, self.q, self.group_size = (x.value for x in v)
        self.outputmode = outputmode
        self.max_trials = DEFAULT_TRIALS if max_trials == 0 else max_trials
        self.max_frames = max_frames
        # counters behind get_average_trials() (reference lib/ldpc_decoder_bb_impl.h:63, .cc:411-419)
        self.total_trials = 0
        self.batch_cnt = 0
        self.frame_cnt = 0

    def close(self):
        if self._h:
            lib.dvbs2_ldpc_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def out_bytes(self):
        return (self.message_bits if self.outputmode == capi.OM_MESSAGE else self.N) // 8

    def _account(self, ret):
        for r in ret:
            self.total_trials += self.max_trials if r < 0 else self.max_trials - int(r)
            self.batch_cnt += 1

    def get_average_trials(self):
        return self.total_trials // self.batch_cnt

    def work(self, llr, want_llr=False):
        """llr: (n_frames, N) int8 host array. Returns (bits (n_frames, out_bytes) uint8, llr_out or None, ret per group)."""
        llr = np.ascontiguousarray(llr, dtype=np.int8)
        nf = llr.shape[0]
        assert llr.shape == (nf, self.N)
        bits = np.empty((nf, self.out_bytes), np.uint8)
        out = np.empty((nf, self.N), np.int8) if want_llr else None
        ng = (nf + self.group_size - 1) // self.group_size
        ret = np.empty(ng, np.int32)
        check(lib.dvbs2_ldpc_decode(self._h, llr.ctypes.data, nf, self.max_trials, self.outputmode,
                                    bits.ctypes.data, out.ctypes.data if want_llr else None, ret.ctypes.data))
        self._account(ret)
        self.frame_cnt += nf
        return bits, out, ret

    def work_device(self, d_llr, n_frames, d_bits, d_llr_out=0, d_ret=0, stream=0):
        """Device-pointer variant (ints from torch.Tensor.data_ptr()); stream = raw hipStream_t value."""
        check(lib.dvbs2_ldpc_decode_device(self._h, d_llr, n_frames, self.max_trials, self.outputmode,
                                           d_bits, d_llr_out or None, d_ret or None, stream or None))

    def enqueue_device(self, d_llr, n_frames, d_bits, d_llr_out=0, d_ret=0, stream=0):
        """work_device without the host synchronisation; finish() completes it."""
        check(lib.dvbs2_ldpc_enqueue_device(self._h, d_llr, n_frames, self.max_trials, self.outputmode,
                                            d_bits, d_llr_out or None, d_ret or None, stream or None))

    def finish(self):
        check(lib.dvbs2_ldpc_finish(self._h))

    @property
    def kernel_name(self):
        return lib.dvbs2_ldpc_kernel_name(self._h).decode()

    def profile(self, enable=True):
        ms, n = C.c_double(), C.c_int()
        check(lib.dvbs2_ldpc_profile(self._h, 1 if enable else 0, ms, n))
        return ms.value, n.value

    @property
    def fallback_rounds(self):
        """host-driven resolution rounds of the group stop since creation (zero in normal operation)"""
        return lib.dvbs2_ldpc_fallback_rounds(self._h)


class BchDecoder:
    """bch_decoder_bb's compute (reference lib/bch_decoder_bb_impl.cc:84-117): n/8-byte codewords -> k/8-byte messages."""

    def __init__(self, standard=capi.STANDARD_DVBS2, framesize=capi.FECFRAME_NORMAL, rate="C1_2", max_frames=64,
                 device=0, raw=None):
        self._h = C.c_void_p()
        if raw is not None:
            m, prim, t, n = raw
            check(lib.dvbs2_bch_create_raw(C.byref(self._h), m, prim, t, n, max_frames, device))
        else:
            check(lib.dvbs2_bch_create(C.byref(self._h), standard, framesize, rate_id(rate), max_frames, device))
        v = [C.c_int() for _ in range(3)]
        check(lib.dvbs2_bch_params(self._h, *v))
        self.n, self.k, self.t = (x.value for x in v)
        # counters behind get_frame_count / get_error_count (lib/bch_decoder_bb_impl.h:46-47)
        self.frame_cnt = 0
        self.frame_error_cnt = 0

    def close(self):
        if self._h:
            lib.dvbs2_bch_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_descramble(self, enable=True):
        """Fuse bbdescrambler_bb (reference lib/bbdescrambler_bb_impl.cc:67-82) into the output stage."""
        check(lib.dvbs2_bch_set_descramble(self._h, int(bool(enable))))

    def genpoly(self):
        g = np.zeros(256, np.uint8)
        deg = check(lib.dvbs2_bch_genpoly(self._h, g.ctypes.data, 256))
        return g[:deg + 1].copy()

    def work(self, cw):
        cw = np.ascontiguousarray(cw, dtype=np.uint8)
        nf = cw.shape[0]
        assert cw.shape == (nf, self.n // 8)
        msg = np.empty((nf, self.k // 8), np.uint8)
        corr = np.empty(nf, np.int32)
        check(lib.dvbs2_bch_decode(self._h, cw.ctypes.data, nf, msg.ctypes.data, corr.ctypes.data))
        self.frame_cnt += nf
        self.frame_error_cnt += int((corr == -1).sum())
        return msg, corr

    def work_device(self, d_cw, n_frames, d_msg, d_corr, stream=0):
        check(lib.dvbs2_bch_decode_device(self._h, d_cw, n_frames, d_msg, d_corr, stream or None))


class Demapper:
    """xfecframe_demapper_cb's compute (reference lib/xfecframe_demapper_cb_impl.cc:101-186)."""

    def __init__(self, framesize=capi.FECFRAME_NORMAL, rate="C1_2", constellation=capi.MOD_QPSK, max_frames=64, device=0):
        self._h = C.c_void_p()
        check(lib.dvbs2_demap_create(C.byref(self._h), framesize, rate_id(rate), constellation, max_frames, device))
        v = [C.c_int() for _ in range(4)]
        check(lib.dvbs2_demap_params(self._h, *v))
        self.n_syms, self.n_llr, self.n_mod, self.column_order = (x.value for x in v)

    def close(self):
        if self._h:
            lib.dvbs2_demap_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def work(self, syms, n0):
        """syms: (n_frames, n_syms) complex64; n0: scalar or (n_frames,) float32 -> (n_frames, n_llr) int8."""
        syms = np.ascontiguousarray(syms, dtype=np.complex64)
        nf = syms.shape[0]
        assert syms.shape == (nf, self.n_syms)
        n0 = np.atleast_1d(np.asarray(n0, np.float32))
        out = np.empty((nf, self.n_llr), np.int8)
        check(lib.dvbs2_demap_soft(self._h, syms.ctypes.data, nf, n0.ctypes.data, len(n0), out.ctypes.data))
        return out

    def estimate_snr(self, syms):
        syms = np.ascontiguousarray(syms, dtype=np.complex64)
        nf = syms.shape[0]
        snr = np.empty(nf, np.float32)
        check(lib.dvbs2_demap_estimate_snr(self._h, syms.ctypes.data, nf, snr.ctypes.data))
        return snr

    def refine_snr(self, syms, ref_llr):
        """Post-decoder linear SNR per frame from the decoded LLRs (handle_llr_pdu, reference :268-307)."""
        syms = np.ascontiguousarray(syms, dtype=np.complex64)
        ref_llr = np.ascontiguousarray(ref_llr, dtype=np.int8)
        nf = syms.shape[0]
        assert ref_llr.shape == (nf, self.n_llr)
        snr = np.empty(nf, np.float32)
        check(lib.dvbs2_demap_refine_snr(self._h, syms.ctypes.data, ref_llr.ctypes.data, nf, snr.ctypes.data))
        return snr

    def work_device(self, d_syms, n_frames, d_n0, n0_count, d_llr, stream=0):
        check(lib.dvbs2_demap_soft_device(self._h, d_syms, n_frames, d_n0, n0_count, d_llr, stream or None))


class PlPayload:
    """PLFRAME payload step of the PL synchroniser (reference lib/plsync_cc_impl.cc:644-653, :727-795): descramble,
    drop the pilot blocks, de-rotate; produces the XFECFRAME symbols the demapper consumes."""

    def __init__(self, gold_code=0, n_slots=360, has_pilots=True, max_frames=16, device=0):
        self._h = C.c_void_p()
        check(lib.dvbs2_plpayload_create(C.byref(self._h), gold_code, n_slots, int(bool(has_pilots)), max_frames, device))
        v = [C.c_int() for _ in range(3)]
        check(lib.dvbs2_plpayload_params(self._h, *v))
        self.payload_len, self.xfecframe_len, self.n_pilots = (x.value for x in v)

    def close(self):
        if self._h:
            lib.dvbs2_plpayload_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def work(self, payload, plheader_phase, fine_foffset, coarse_corrected, pilot_phase=None):
        """payload: (n_frames, payload_len) complex64; per-frame arrays as in plframe_info_t / pl_freq_sync."""
        payload = np.ascontiguousarray(payload, dtype=np.complex64)
        nf = payload.shape[0]
        assert payload.shape == (nf, self.payload_len)
        hph = np.ascontiguousarray(plheader_phase, np.float32)
        cc = np.ascontiguousarray(coarse_corrected, np.int32)
        inc = np.ascontiguousarray(2.0 * np.pi * np.asarray(fine_foffset, np.float64), np.float32)  # :730-731
        pp = np.ascontiguousarray(pilot_phase if pilot_phase is not None else np.zeros((nf, max(self.n_pilots, 1))), np.float32)
        out = np.empty((nf, self.xfecframe_len), np.complex64)
        check(lib.dvbs2_plpayload_process(self._h, payload.ctypes.data, nf, hph.ctypes.data, inc.ctypes.data, cc.ctypes.data,
                                          pp.ctypes.data, out.ctypes.data))
        return out


def pl_scrambling_rn(gold_code, n):
    rn = np.zeros(n, np.uint8)
    check(lib.dvbs2_pl_scrambling_rn(gold_code, rn.ctypes.data, n))
    return rn


class BbDeheader:
    """bbdeheader_bb (reference lib/bbdeheader_bb_impl.cc): descrambled BBFRAMEs in, 188-byte MPEG-TS packets out. The block's
    state (synchronised flag, partial packet, counters) is kept in the handle between calls, as between work() calls."""

    def __init__(self, standard=capi.STANDARD_DVBS2, framesize=capi.FECFRAME_NORMAL, rate="C1_2", max_frames=64, device=0,
                 kbch_bits=None):
        self._h = C.c_void_p()
        if kbch_bits is not None:
            check(lib.dvbs2_bbdeheader_create_raw(C.byref(self._h), kbch_bits, max_frames, device))
        else:
            check(lib.dvbs2_bbdeheader_create(C.byref(self._h), standard, framesize, rate_id(rate), max_frames, device))
        v = [C.c_int() for _ in range(3)]
        check(lib.dvbs2_bbdeheader_params(self._h, *v))
        self.kbch_bytes, self.max_dfl, self.max_out_bytes_per_frame = (x.value for x in v)

    def close(self):
        if self._h:
            lib.dvbs2_bbdeheader_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def work(self, bbframes):
        """bbframes: (n_frames, kbch_bytes) uint8 -> the TS bytes produced (general_work's output items)."""
        bb = np.ascontiguousarray(bbframes, dtype=np.uint8)
        nf = bb.shape[0] if bb.ndim == 2 else bb.size // self.kbch_bytes
        assert bb.size == nf * self.kbch_bytes
        out = np.empty(max(nf, 1) * self.max_out_bytes_per_frame, np.uint8)
        produced = C.c_int64()
        check(lib.dvbs2_bbdeheader_process(self._h, bb.ctypes.data, nf, out.ctypes.data, C.byref(produced)))
        return out[:produced.value].copy()

    def work_device(self, d_bbframes, n_frames, d_ts_out, stream=0):
        check(lib.dvbs2_bbdeheader_process_device(self._h, d_bbframes, n_frames, d_ts_out, stream))

    def finish(self, stream=0):
        produced = C.c_int64()
        check(lib.dvbs2_bbdeheader_finish(self._h, C.byref(produced), stream))
        return produced.value

    def counters(self, stream=0):
        c = capi.BbDeheaderCounters()
        check(lib.dvbs2_bbdeheader_counters(self._h, C.byref(c), stream))
        return {k: getattr(c, k) for k, _ in c._fields_}

    def reset(self, stream=0):
        check(lib.dvbs2_bbdeheader_reset(self._h, stream))


class FecChain:
    """demapper -> LDPC (OM_MESSAGE) -> BCH on the device, as wired in apps/dvbs2-rx:853-863."""

    def __init__(self, standard=capi.STANDARD_DVBS2, framesize=capi.FECFRAME_NORMAL, rate="C3_4",
                 constellation=capi.MOD_8PSK, group_size=32, max_frames=64, max_trials=0, device=0, from_llr=False):
        self._h = C.c_void_p()
        if from_llr:  # ldpc_decoder_bb -> bch_decoder_bb only (LLRs in)
            check(lib.dvbs2_chain_create_llr(C.byref(self._h), standard, framesize, rate_id(rate), group_size, max_frames, device))
        else:
            check(lib.dvbs2_chain_create(C.byref(self._h), standard, framesize, rate_id(rate), constellation,
                                         group_size, max_frames, device))
        a, b = C.c_int(), C.c_int()
        check(lib.dvbs2_chain_params(self._h, a, b))
        self.n_syms, self.msg_bytes = a.value, b.value
        check(lib.dvbs2_chain_llr_params(self._h, a, b, None))
        self.n_llr = a.value
        self.group_size = group_size
        self.max_trials = DEFAULT_TRIALS if max_trials == 0 else max_trials

    def close(self):
        if self._h:
            lib.dvbs2_chain_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_descramble(self, enable=True):
        check(lib.dvbs2_chain_set_descramble(self._h, int(bool(enable))))

    def work(self, syms, n0, want_ret=True):
        """HOST buffers (dvbs2_chain_decode): syms complex64 [n_frames, n_syms] (or float32 [n_frames, 2 n_syms]), n0 scalar or
        one float per frame. Returns (msg uint8 [n_frames, msg_bytes], ldpc_ret int32 per group, bch_corr int32 per frame)."""
        syms = np.ascontiguousarray(syms)
        n_frames = syms.shape[0]
        n0 = np.ascontiguousarray(np.atleast_1d(np.asarray(n0, np.float32)))
        msg = np.empty((n_frames, self.msg_bytes), np.uint8)
        ret = np.empty(((n_frames + self.group_size - 1) // self.group_size,), np.int32)
        corr = np.empty((n_frames,), np.int32)
        check(lib.dvbs2_chain_decode(self._h, syms.ctypes.data, n_frames, n0.ctypes.data, int(n0.size), self.max_trials,
                                     msg.ctypes.data, ret.ctypes.data if want_ret else None, corr.ctypes.data if want_ret else None))
        return msg, ret, corr

    def work_llr(self, llr, want_ret=True):
        """HOST buffers (dvbs2_chain_decode_llr): llr int8 [n_frames, N]."""
        llr = np.ascontiguousarray(llr, np.int8)
        n_frames = llr.shape[0]
        msg = np.empty((n_frames, self.msg_bytes), np.uint8)
        ret = np.empty(((n_frames + self.group_size - 1) // self.group_size,), np.int32)
        corr = np.empty((n_frames,), np.int32)
        check(lib.dvbs2_chain_decode_llr(self._h, llr.ctypes.data, n_frames, self.max_trials, msg.ctypes.data,
                                         ret.ctypes.data if want_ret else None, corr.ctypes.data if want_ret else None))
        return msg, ret, corr

    def work_host_ptr(self, syms_ptr, n_frames, n0_ptr, n0_count, msg_ptr, ret_ptr=0, corr_ptr=0):
        """dvbs2_chain_decode on raw HOST addresses (page-locked buffers of the caller: bench.py)."""
        check(lib.dvbs2_chain_decode(self._h, syms_ptr, n_frames, n0_ptr, n0_count, self.max_trials, msg_ptr, ret_ptr or None, corr_ptr or None))

    def work_device(self, d_syms, n_frames, d_n0, n0_count, d_msg, d_ldpc_ret=0, d_bch_corr=0, stream=0):
        check(lib.dvbs2_chain_decode_device(self._h, d_syms, n_frames, d_n0, n0_count, self.max_trials, d_msg,
                                            d_ldpc_ret or None, d_bch_corr or None, stream or None))

    def work_llr_device(self, d_llr, n_frames, d_msg, d_ldpc_ret=0, d_bch_corr=0, stream=0):
        check(lib.dvbs2_chain_decode_llr_device(self._h, d_llr, n_frames, self.max_trials, d_msg,
                                                d_ldpc_ret or None, d_bch_corr or None, stream or None))

    def enqueue_device(self, d_syms, n_frames, d_n0, n0_count, d_msg, d_ldpc_ret=0, d_bch_corr=0, stream=0):
        check(lib.dvbs2_chain_enqueue_device(self._h, d_syms, n_frames, d_n0, n0_count, self.max_trials, d_msg,
                                             d_ldpc_ret or None, d_bch_corr or None, stream or None))

    def enqueue_llr_device(self, d_llr, n_frames, d_msg, d_ldpc_ret=0, d_bch_corr=0, stream=0):
        check(lib.dvbs2_chain_enqueue_llr_device(self._h, d_llr, n_frames, self.max_trials, d_msg,
                                                 d_ldpc_ret or None, d_bch_corr or None, stream or None))

    def finish(self):
        check(lib.dvbs2_chain_finish(self._h))

    @property
    def kernel_name(self):
        return lib.dvbs2_chain_ldpc_kernel_name(self._h).decode()

    def profile(self, enable=True):
        ms, n = C.c_double(), C.c_int()
        check(lib.dvbs2_chain_ldpc_profile(self._h, 1 if enable else 0, ms, n))
        return ms.value, n.value
