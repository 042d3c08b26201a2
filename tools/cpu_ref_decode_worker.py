#!/usr/bin/env python3
"""One CPU worker of the full-batch parity checks (test infrastructure: it drives the CHECKER, never the product path).
Decodes groups [g0, g1) of an int8 LLR batch IN PLACE in a shared .npy file with the genuine reference decoder
(oracle/_ref; impl 0 = AVX2, 32 frames per call; 2 = generic, 16) and writes the per-group return values. One PROCESS per
worker: the reference keeps one global decoder object per translation unit (lib/ldpc_decoder/ldpc_decoder_avx2.cc:21).
usage: cpu_ref_decode_worker.py table impl trials llr.npy ret.npy g0 g1"""
import ctypes as C, os, sys
import numpy as np
table, impl, trials, f_llr, f_ret, g0, g1 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5], int(sys.argv[6]), int(sys.argv[7])
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
r = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libdvbs2_ref_ldpc.so"))
r.ref_ldpc_init.argtypes = [C.c_char_p, C.c_int]; r.ref_ldpc_decode.argtypes = [C.c_void_p, C.c_int]
G = r.ref_ldpc_init(table.encode(), impl)
assert G > 0, table
llr = np.load(f_llr, mmap_mode="r+")
ret = np.load(f_ret, mmap_mode="r+")
for g in range(g0, g1):
    blk = np.ascontiguousarray(llr[g * G:(g + 1) * G])  # private copy: the decoder works in place
    ret[g] = r.ref_ldpc_decode(blk.ctypes.data_as(C.c_void_p), trials)
    llr[g * G:(g + 1) * G] = blk
llr.flush(); ret.flush()
