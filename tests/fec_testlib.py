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
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("ldpc_oracle.c", "bch_oracle.c", "demap_oracle.c")]
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
