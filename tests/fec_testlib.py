"""Shared helpers for the tests: loads the CPU checkers (oracle/) and builds deterministic inputs.

Only test code (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline leg) may import this
module: it is the bridge to oracle/ and is never used by the product path.
"""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")


def _build_oracle():
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("ldpc_oracle.c", "bch_oracle.c", "demap_oracle.c", "bb_oracle.c", "bbdeheader_oracle.c", "pl_oracle.c")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


_oracle = None


def oracle():
    """ctypes handle of oracle/liboracle.so (the plain-C restatement)."""
    global _oracle
    if _oracle is None:
        o = C.CDLL(_build_oracle())
        o.oracle_ldpc_decode.argtypes = [C.c_char_p, C.c_int, C.c_void_p, C.c_int]
        o.oracle_ldpc_decode.restype = C.c_int
        o.oracle_ldpc_pack.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
        o.oracle_ldpc_encode.argtypes = [C.c_char_p, C.c_void_p, C.c_void_p]
        o.oracle_ldpc_info.argtypes = [C.c_char_p] + [C.POINTER(C.c_int)] * 4
        o.oracle_bch_new.argtypes = [C.c_int, C.c_uint32, C.c_int, C.c_int]
        o.oracle_bch_new.restype = C.c_void_p
        o.oracle_bch_free.argtypes = [C.c_void_p]
        for f in ("oracle_bch_k", "oracle_bch_n", "oracle_bch_gdeg"):
            getattr(o, f).argtypes = [C.c_void_p]
        o.oracle_bch_genpoly.argtypes = [C.c_void_p, C.c_void_p]
        o.oracle_bch_alpha.argtypes = [C.c_void_p, C.c_int]
        o.oracle_bch_alpha.restype = C.c_uint32
        o.oracle_bch_minpoly.argtypes = [C.c_void_p, C.c_int]
        o.oracle_bch_minpoly.restype = C.c_uint32
        o.oracle_bch_encode_bits.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        o.oracle_bch_syndrome_bits.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        o.oracle_bch_err_loc_poly.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        o.oracle_bch_err_loc_numbers.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        o.oracle_bch_decode_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        o.oracle_bch_encode_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        o.oracle_demap_qpsk.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_void_p]
        o.oracle_demap_8psk.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
        o.oracle_demap_snr.argtypes = [C.c_void_p, C.c_int, C.c_int]
        o.oracle_demap_snr.restype = C.c_float
        o.oracle_demap_snr_refined.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int]
        o.oracle_demap_snr_refined.restype = C.c_float
        o.oracle_bb_sequence.argtypes = [C.c_void_p, C.c_int]
        o.oracle_pl_rn.argtypes = [C.c_int, C.c_void_p, C.c_int]
        o.oracle_pl_payload.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p]
        o.oracle_bb_descramble.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        o.oracle_crc8_rem.argtypes = [C.c_void_p, C.c_int]
        o.oracle_crc8_rem.restype = C.c_uint8
        o.oracle_bbdh_init.argtypes = [C.c_void_p, C.c_int]
        o.oracle_bbdh_work.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        o.oracle_bbdh_work.restype = C.c_longlong
        _oracle = o
    return _oracle


_ref = None


def ref_ldpc():
    """ctypes handle of oracle/_ref/libdvbs2_ref_ldpc.so (the genuine reference LDPC decoders), or None."""
    global _ref
    if _ref is None:
        p = os.path.join(ORACLE_DIR, "_ref", "libdvbs2_ref_ldpc.so")
        if not os.path.exists(p):
            return None
        r = C.CDLL(p)
        r.ref_ldpc_init.argtypes = [C.c_char_p, C.c_int]
        r.ref_ldpc_decode.argtypes = [C.c_void_p, C.c_int]
        _ref = r
    return _ref


_ref_bch = None


def ref_bch():
    """ctypes handle of oracle/_ref/libdvbs2_ref_bch.so (the genuine reference BCH codec), or None."""
    global _ref_bch
    if _ref_bch is None:
        p = os.path.join(ORACLE_DIR, "_ref", "libdvbs2_ref_bch.so")
        if not os.path.exists(p):
            return None
        r = C.CDLL(p)
        r.ref_bch_new.argtypes = [C.c_uint32, C.c_int, C.c_int]
        r.ref_bch_new.restype = C.c_void_p
        r.ref_bch_free.argtypes = [C.c_void_p]
        r.ref_bch_k.argtypes = [C.c_void_p]
        r.ref_bch_n.argtypes = [C.c_void_p]
        r.ref_bch_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        r.ref_bch_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        r.ref_crc8_rem.argtypes = [C.c_void_p, C.c_int]
        _ref_bch = r
    return _ref_bch


class RefBch:
    """The genuine reference bch_codec<uint32_t, bitset256_t> (container build, oracle/_ref). decode() returns the
    reference's return value, or -2 where it throws."""

    def __init__(self, prim_poly, t, n=0):
        self.r = ref_bch()
        assert self.r is not None, "oracle/_ref/libdvbs2_ref_bch.so absent"
        self.h = self.r.ref_bch_new(prim_poly, t, n)
        assert self.h, "reference constructor threw"
        self.n, self.k = self.r.ref_bch_n(self.h), self.r.ref_bch_k(self.h)

    def decode(self, cw):
        cw = np.ascontiguousarray(cw, np.uint8).reshape(-1, self.n // 8)
        msg = np.zeros((cw.shape[0], self.k // 8), np.uint8)
        ret = [self.r.ref_bch_decode(self.h, ptr(cw[f]), ptr(msg[f])) for f in range(cw.shape[0])]
        return msg, ret

    def encode(self, msg):
        msg = np.ascontiguousarray(msg, np.uint8).reshape(-1, self.k // 8)
        cw = np.zeros((msg.shape[0], self.n // 8), np.uint8)
        for f in range(msg.shape[0]):
            self.r.ref_bch_encode(self.h, ptr(msg[f]), ptr(cw[f]))
        return cw

    def close(self):
        if self.h:
            self.r.ref_bch_free(self.h)
            self.h = None


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def ldpc_info(table):
    n, k, q, lt = C.c_int(), C.c_int(), C.c_int(), C.c_int()
    assert oracle().oracle_ldpc_info(table.encode(), n, k, q, lt) == 0, table
    return n.value, k.value, q.value, lt.value


def oracle_ldpc_decode(table, llr, G, trials):
    """llr: (n_frames, N) int8, n_frames % G == 0. Returns (decoded llr copy, list of return values per group)."""
    out = np.ascontiguousarray(llr).copy()
    rets = []
    for g in range(0, out.shape[0], G):
        blk = out[g:g + G]
        rets.append(oracle().oracle_ldpc_decode(table.encode(), G, ptr(blk), trials))
    return out, rets


def ref_ldpc_decode(table, llr, impl, trials):
    """Genuine reference. impl 0 = AVX2 (G=32), 1 = SSE4.1 (G=16), 2 = generic (G=16)."""
    r = ref_ldpc()
    G = r.ref_ldpc_init(table.encode(), impl)
    assert G > 0
    out = np.ascontiguousarray(llr).copy()
    rets = []
    for g in range(0, out.shape[0], G):
        blk = out[g:g + G]
        rets.append(r.ref_ldpc_decode(ptr(blk), trials))
    return out, rets


def cpu_ldpc_decode_ragged(table, llr, G, trials):
    """Any number of frames: whole groups of G through the genuine reference when it serves that G (32: AVX2), the trailing partial
    group -- a group of its own, like the HIP path treats it -- through the scalar restatement (which takes any group size)."""
    nf = llr.shape[0]
    full = nf - nf % G
    outs, rets = [], []
    if full:
        if ref_ldpc() is not None and G == 32:
            o, r = ref_ldpc_decode(table, llr[:full], 0, trials)
        else:
            o, r = oracle_ldpc_decode(table, llr[:full], G, trials)
        outs.append(o); rets += list(r)
    if nf > full:
        o, r = oracle_ldpc_decode(table, llr[full:], nf - full, trials)
        outs.append(o); rets += list(r)
    return np.concatenate(outs), rets


def ref_ldpc_decode_parallel(table, llr, impl, trials, procs=None):
    """Genuine reference on a WHOLE batch: worker processes (tools/cpu_ref_decode_worker.py), each decoding a contiguous
    range of groups of a shared .npy file in /dev/shm. Returns (decoded llr, list of return values per group), identical
    to ref_ldpc_decode(). n_frames must be a multiple of the reference's batch (32 for AVX2, 16 otherwise)."""
    import sys
    import tempfile
    G = 32 if impl == 0 else 16
    nf = llr.shape[0]
    assert nf % G == 0, "whole reference batches only"
    ng = nf // G
    if procs is None:
        procs = max(1, min(len(os.sched_getaffinity(0)), 64, ng))
    if procs == 1 or ng < 4:
        return ref_ldpc_decode(table, llr, impl, trials)
    tmp = "/dev/shm" if os.path.isdir("/dev/shm") else None
    with tempfile.TemporaryDirectory(dir=tmp) as d:
        f_llr, f_ret = os.path.join(d, "llr.npy"), os.path.join(d, "ret.npy")
        np.save(f_llr, np.ascontiguousarray(llr, np.int8))
        np.save(f_ret, np.zeros(ng, np.int32))
        worker = os.path.join(ROOT, "tools", "cpu_ref_decode_worker.py")
        cuts = [ng * i // procs for i in range(procs + 1)]
        ps = [subprocess.Popen([sys.executable, worker, table, str(impl), str(trials), f_llr, f_ret, str(cuts[i]), str(cuts[i + 1])])
              for i in range(procs) if cuts[i + 1] > cuts[i]]
        for p in ps:
            assert p.wait() == 0, "reference worker failed"
        return np.load(f_llr), np.load(f_ret).tolist()


def pack_bits(llr, nbits):
    """Hard decision + MSB-first packing (lib/ldpc_decoder_bb_impl.cc:432-442)."""
    out = np.zeros((llr.shape[0], nbits // 8), np.uint8)
    oracle().oracle_ldpc_pack(ptr(np.ascontiguousarray(llr)), llr.shape[0], llr.shape[1], nbits, ptr(out))
    return out


# ------------------------------------------------------------------ input generators
def llr_noise(n_frames, N, seed, sigma=8.0):
    """Never-converging input of SURVEY 8(d): i.i.d. clamp(round(N(0, sigma^2)))."""
    rng = np.random.default_rng(seed)
    return np.clip(np.rint(rng.normal(0.0, sigma, (n_frames, N))), -128, 127).astype(np.int8)


def ldpc_encode(table, info_bits):
    """info_bits: (n_frames, K) uint8 in {0,1} -> codeword bits (n_frames, N)."""
    N, K, _, _ = ldpc_info(table)
    cw = np.zeros((info_bits.shape[0], N), np.uint8)
    for f in range(info_bits.shape[0]):
        assert oracle().oracle_ldpc_encode(table.encode(), ptr(np.ascontiguousarray(info_bits[f])), ptr(cw[f])) == 0
    return cw


def llr_codeword_awgn(table, n_frames, seed, amp=6.0, sigma=4.0, info=None):
    """Valid codewords through a BPSK-like AWGN channel quantised to int8 (bit 0 -> +amp)."""
    N, K, _, _ = ldpc_info(table)
    rng = np.random.default_rng(seed)
    if info is None:
        info = rng.integers(0, 2, (n_frames, K), dtype=np.uint8)
    cw = ldpc_encode(table, info)
    y = amp * (1.0 - 2.0 * cw) + sigma * rng.normal(0.0, 1.0, cw.shape)
    return np.clip(np.rint(y), -128, 127).astype(np.int8), cw


def make_input(table, kind, n_frames, seed=0, amp=6.0, sigma=4.0):
    """Deterministic LDPC test inputs by name (used by the golden fixtures)."""
    N = ldpc_info(table)[0]
    if kind == "noise":
        return llr_noise(n_frames, N, seed)
    if kind == "awgn":
        return llr_codeword_awgn(table, n_frames, seed, amp=amp, sigma=sigma)[0]
    if kind == "sat":
        rng = np.random.default_rng(seed)
        return rng.choice(np.array([-128, -127, 127, 126, 0], np.int8), (n_frames, N))
    if kind == "zero":
        return np.zeros((n_frames, N), np.int8)
    if kind == "clean":  # valid codewords with strong LLRs: the first syndrome test passes, zero updates
        return llr_codeword_awgn(table, n_frames, seed, amp=20, sigma=0.0)[0]
    raise ValueError(kind)


# ------------------------------------------------------------------ BCH helpers (oracle)
class OracleBch:
    def __init__(self, m, prim_poly, t, n=0):
        self.o = oracle()
        self.h = self.o.oracle_bch_new(m, prim_poly, t, n)
        self.n, self.k, self.t, self.m = self.o.oracle_bch_n(self.h), self.o.oracle_bch_k(self.h), t, m

    def __del__(self):
        try:
            self.o.oracle_bch_free(self.h)
        except Exception:
            pass

    def genpoly_int(self):
        deg = self.o.oracle_bch_gdeg(self.h)
        g = np.zeros(deg + 1, np.uint8)
        self.o.oracle_bch_genpoly(self.h, ptr(g))
        return sum(int(b) << i for i, b in enumerate(g))

    def alpha(self, i):
        return self.o.oracle_bch_alpha(self.h, i)

    def minpoly(self, e):
        return self.o.oracle_bch_minpoly(self.h, e)

    def encode_bits(self, msg_bits):
        cw = np.zeros(self.n, np.uint8)
        self.o.oracle_bch_encode_bits(self.h, ptr(np.ascontiguousarray(msg_bits, np.uint8)), ptr(cw))
        return cw

    def syndrome_bits(self, cw_bits):
        S = np.zeros(2 * self.t, np.uint32)
        n = self.o.oracle_bch_syndrome_bits(self.h, ptr(np.ascontiguousarray(cw_bits, np.uint8)), ptr(S))
        return S[:n]

    def err_loc(self, S):
        sigma = np.zeros(64, np.uint32)
        deg = self.o.oracle_bch_err_loc_poly(self.h, ptr(np.ascontiguousarray(S, np.uint32)), ptr(sigma))
        nums = np.zeros(32, np.uint32)
        cnt = self.o.oracle_bch_err_loc_numbers(self.h, ptr(sigma), deg, ptr(nums))
        return sigma[:deg + 1], nums[:max(cnt, 0)], cnt

    def encode_bytes(self, msg):
        msg = np.ascontiguousarray(msg, np.uint8)
        cw = np.zeros((msg.shape[0], self.n // 8), np.uint8)
        for f in range(msg.shape[0]):
            self.o.oracle_bch_encode_bytes(self.h, ptr(msg[f]), ptr(cw[f]))
        return cw

    def decode_bytes(self, cw):
        cw = np.ascontiguousarray(cw, np.uint8)
        msg = np.zeros((cw.shape[0], self.k // 8), np.uint8)
        ret = np.zeros(cw.shape[0], np.int32)
        for f in range(cw.shape[0]):
            ret[f] = self.o.oracle_bch_decode_bytes(self.h, ptr(cw[f]), ptr(msg[f]))
        return msg, ret


def bch_golden_input(codec, n, k, case):
    """Received word of one tests/golden/bch_golden.json case. codec: anything with encode_bytes()/encode() (the
    systematic encoder is pinned by the reference's own KATs and by the committed sha_in)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bch_craft
    if "exps" in case:
        return bch_craft.word_from_exponents(n, case["exps"])
    if case.get("garbage"):
        return np.random.default_rng(case["seed"]).integers(0, 256, n // 8, dtype=np.uint8)
    msg = np.random.default_rng(case["seed"]).integers(0, 256, (1, k // 8), dtype=np.uint8)
    cw = codec.encode(msg)[0] if hasattr(codec, "encode") else codec.encode_bytes(msg)[0]
    return flip_bits(cw, case["flips"])


def chain_expect(table, bch_n, bch_t, framesize, llr, trials):
    """What ldpc_decoder_bb (OM_MESSAGE) -> bch_decoder_bb give for a WHOLE batch of int8 LLR frames, G = 32: the genuine
    AVX2 LDPC reference on all cores (the restatement without oracle/_ref), then the BCH codec frame by frame (the genuine
    bch.cc when oracle/_ref holds it, else the restatement). Returns (messages, corrections per frame, LDPC return values,
    who)."""
    if ref_ldpc() is not None:
        dec_llr, wret = ref_ldpc_decode_parallel(table, llr, 0, trials); who = "reference AVX2 LDPC"
    else:
        dec_llr, wret = oracle_ldpc_decode(table, llr, 32, trials); who = "oracle LDPC"
    cw = pack_bits(dec_llr, bch_n)
    m, prim = BCH_FIELDS[framesize]
    if ref_bch() is not None:
        rb = RefBch(prim, bch_t, bch_n)
        msg, corr = rb.decode(cw)
        rb.close()
        return msg, np.asarray(corr, np.int32), wret, who + " + reference BCH"
    msg, corr = OracleBch(m, prim, bch_t, bch_n).decode_bytes(cw)
    return msg, corr, wret, who + " + BCH oracle"


BCH_FIELDS = {1: (16, 0b10000000000101101), 0: (14, 0b100000000101011), 2: (15, 0b1000000000101101)}  # by framesize id


def flip_bits(cw_bytes, positions):
    """Flip stream bit positions (0 = first transmitted bit) in a packed byte row (copy)."""
    out = cw_bytes.copy()
    for p in positions:
        out[p // 8] ^= np.uint8(1 << (7 - p % 8))
    return out


# ------------------------------------------------------------------ demapper helpers (oracle)
def oracle_demap(syms, n0, constellation, order=0):
    syms = np.ascontiguousarray(syms, np.complex64)
    nf, ns = syms.shape
    n0 = np.broadcast_to(np.asarray(n0, np.float32), (nf,))
    nmod = 2 if constellation == 4 else 3
    out = np.zeros((nf, ns * nmod), np.int8)
    for f in range(nf):
        if constellation == 4:
            oracle().oracle_demap_qpsk(ptr(syms[f]), ns, float(n0[f]), ptr(out[f]))
        else:
            oracle().oracle_demap_8psk(ptr(syms[f]), ns, float(n0[f]), order, ptr(out[f]))
    return out


M8PSK = np.array([np.sqrt(.5) * (1 + 1j), 1, -1, np.sqrt(.5) * (-1 - 1j), 1j, np.sqrt(.5) * (1 - 1j),
                  np.sqrt(.5) * (-1 + 1j), -1j], np.complex64)


def map_8psk(bits3):
    """bits3: (..., 3) bits (b0 b1 b2 of lib/psk.hh:152-157 in +-1 -> index form) -> constellation point."""
    b = 1 - 2 * bits3.astype(np.int32)  # bit 0 -> +1, bit 1 -> -1 (positive LLR = bit 0)
    idx = (((b[..., 0] + 1) << 1) ^ 0x4) | ((b[..., 1] + 1) ^ 0x2) | (((b[..., 2] + 1) >> 1) ^ 0x1)
    return M8PSK[idx]


def oracle_bb_descramble(msg):
    """msg: (n_frames, kbch_bytes) uint8 -> descrambled copy (lib/bbdescrambler_bb_impl.cc:67-82)."""
    msg = np.ascontiguousarray(msg, np.uint8)
    out = np.empty_like(msg)
    oracle().oracle_bb_descramble(ptr(msg), ptr(out), msg.shape[1], msg.shape[0])
    return out


def oracle_pl_payload(payload, n_slots, has_pilots, gold, plheader_phase, fine_foffset, coarse, pilot_phase):
    """payload: (n_frames, payload_len) complex64 -> (n_frames, 90 n_slots) complex64 (lib/plsync_cc_impl.cc:644-795)."""
    payload = np.ascontiguousarray(payload, np.complex64)
    nf = payload.shape[0]
    out = np.empty((nf, n_slots * 90), np.complex64)
    pp = np.ascontiguousarray(pilot_phase, np.float32)
    for f in range(nf):
        oracle().oracle_pl_payload(ptr(payload[f]), n_slots, int(has_pilots), gold, float(plheader_phase[f]), float(fine_foffset[f]),
                                   int(coarse[f]), ptr(np.ascontiguousarray(pp[f])), ptr(out[f]))
    return out


# ---- BBFRAME de-header (oracle/bbdeheader_oracle.c) -------------------------------------------------------------------
class _BbdhOracleState(C.Structure):
    _fields_ = [("kbch_bytes", C.c_int), ("max_dfl", C.c_int), ("synched", C.c_int), ("partial", C.c_int),
                ("partial_pkt", C.c_uint8 * 188), ("packets", C.c_uint64), ("errors", C.c_uint64), ("bbframes", C.c_uint64),
                ("dropped", C.c_uint64), ("gaps", C.c_uint64), ("overruns", C.c_uint64)]


class OracleBbDeheader:
    """The plain-C restatement of bbdeheader_bb (stateful like the block)."""

    def __init__(self, kbch_bits):
        self.s = _BbdhOracleState()
        oracle().oracle_bbdh_init(C.byref(self.s), kbch_bits)
        self.kbch_bytes = kbch_bits // 8
        self.max_out = ((kbch_bits - 80) // 8 + 187) // 188 * 188

    def work(self, bbframes):
        bb = np.ascontiguousarray(bbframes, np.uint8)
        nf = bb.size // self.kbch_bytes
        out = np.empty(max(nf, 1) * self.max_out, np.uint8)
        n = oracle().oracle_bbdh_work(C.byref(self.s), ptr(bb), nf, ptr(out))
        return out[:n].copy()

    def counters(self):
        s = self.s
        return dict(packets=s.packets, errors=s.errors, bbframes=s.bbframes, dropped=s.dropped, gaps=s.gaps, overruns=s.overruns,
                    synched=s.synched, partial_ts_bytes=s.partial)


def crc8_dvbs2(data):
    """CRC-8 of DVB-S2 (x^8 + x^7 + x^6 + x^4 + x^2 + 1, zero start, no reflection) of a bytes-like, bit-serial."""
    reg = 0
    for byte in bytes(data):
        for b in range(7, -1, -1):
            reg = (reg << 1) | ((byte >> b) & 1)
            if reg & 0x100:
                reg ^= 0x1D5
    for _ in range(8):  # append eight zero bits: remainder of data * x^8
        reg <<= 1
        if reg & 0x100:
            reg ^= 0x1D5
    return reg & 0xff


def ts_up_stream(n_ups, rng):
    """n_ups MPEG-TS user packets: sync byte, three zero bytes, 184 random bytes (as the reference's test builds them)."""
    ups = rng.integers(0, 256, (n_ups, 188), dtype=np.uint8)
    ups[:, 0] = 0x47
    ups[:, 1:4] = 0
    return ups.reshape(-1)


def ts_crc_encode(up_stream):
    """Mode adaptation: the sync byte of every user packet but the first is replaced by the CRC-8 of the preceding packet's
    187 bytes after its sync byte (EN 302 307-1 clause 5.1.4)."""
    s = np.array(up_stream, np.uint8).copy()
    n = s.size // 188
    for i in range(1, n):
        s[i * 188] = crc8_dvbs2(s[(i - 1) * 188 + 1:i * 188])
    return s


def bbheader(kbch_bits, syncd_bits, dfl_bits=None, upl_bits=188 * 8, matype1=0xF2, matype2=0, sync=0x47):
    """Ten BBHEADER bytes with a correct CRC-8: MATYPE (TS, SIS, CCM, roll-off 0.2), UPL, DFL, SYNC, SYNCD."""
    if dfl_bits is None:
        dfl_bits = kbch_bits - 80
    h = bytes([matype1, matype2, upl_bits >> 8, upl_bits & 0xff, dfl_bits >> 8, dfl_bits & 0xff, sync, syncd_bits >> 8, syncd_bits & 0xff])
    return np.frombuffer(h + bytes([crc8_dvbs2(h)]), np.uint8)


def bbframe_stream(kbch_bits, n_frames, up_stream, syncd_bits=0):
    """n_frames BBFRAMEs whose DATAFIELDs carry the CRC-encoded user packet stream back to back (maximum DFL); SYNCD follows
    from where the first packet of each DATAFIELD starts."""
    dfl_bytes = (kbch_bits - 80) // 8
    enc = ts_crc_encode(up_stream)
    assert enc.size >= n_frames * dfl_bytes
    frames = np.zeros((n_frames, kbch_bits // 8), np.uint8)
    off = 0
    for i in range(n_frames):
        frames[i, :10] = bbheader(kbch_bits, syncd_bits)
        frames[i, 10:] = enc[off:off + dfl_bytes]
        partial = (off + dfl_bytes) % 188
        syncd_bits = ((188 - partial) % 188) * 8  # (a packet that starts with the DATAFIELD: SYNCD = 0, EN 302 307-1 clause 5.1.6)
        off += dfl_bytes
    return frames
