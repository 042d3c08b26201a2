#!/usr/bin/env python3
"""One CPU worker of bench.py's cpu_baseline leg (test infrastructure: it drives the CHECKER, never the product path).
Pins itself to one CPU, runs the genuine reference AVX2 LDPC decoder (oracle/_ref, impl 0, 32 frames per call -- the
reference's unit of work, lib/ldpc_decoder_bb_impl.cc:406-410; its decoder object is a global per translation unit,
lib/ldpc_decoder/ldpc_decoder_avx2.cc:21, hence one PROCESS per core) on noise LLRs from a common start time for a fixed
wall time and prints "frames seconds".  usage: cpu_ref_worker.py table trials seconds cpu start_epoch"""
import ctypes as C, os, sys, time
import numpy as np
table, trials, seconds, cpu, start = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5])
try:
    os.sched_setaffinity(0, {cpu})
except OSError:
    pass
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
r = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libdvbs2_ref_ldpc.so"))
r.ref_ldpc_init.argtypes = [C.c_char_p, C.c_int]; r.ref_ldpc_decode.argtypes = [C.c_void_p, C.c_int]
G = r.ref_ldpc_init(table.encode(), 0)
N = int(sys.argv[6])
rng = np.random.default_rng(3000 + cpu)
xs = [np.clip(np.rint(rng.normal(0.0, 8.0, (G, N))), -128, 127).astype(np.int8) for _ in range(2)]
while time.time() < start:
    time.sleep(0.001)
t0 = time.perf_counter(); frames = 0; i = 0
while time.perf_counter() - t0 < seconds:
    y = xs[i & 1].copy(); i += 1
    r.ref_ldpc_decode(y.ctypes.data_as(C.c_void_p), trials)
    frames += G
print(frames, time.perf_counter() - t0)
