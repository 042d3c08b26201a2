#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point dvbs2_ldpc_decode (what a GNU Radio block would call):
pageable numpy buffers in, packed bits out, table B4, 4096 frames, cap 50, noise input."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-dvbs2rx_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fec_testlib as T
from dvbs2rx_amd import LdpcDecoder, capi
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
pinned = len(sys.argv) > 2 and sys.argv[2] == "pinned"  # the block page-locked its buffer once (dvbs2_host_register)
N = 64800
llr = T.llr_noise(nf, N, 1)
dec = LdpcDecoder(standard=capi.STANDARD_DVBS2, framesize=capi.FECFRAME_NORMAL, rate="C1_2", outputmode=capi.OM_MESSAGE,
                  max_trials=50, group_size=32, max_frames=nf)
if pinned:
    capi.check(capi.lib.dvbs2_host_register(llr.ctypes.data, llr.nbytes))
for _ in range(2):
    dec.work(llr)
t0 = time.perf_counter()
for _ in range(3):
    bits, _, ret = dec.work(llr)
dt = (time.perf_counter() - t0) / 3
print(f"host-buffer API ({'page-locked' if pinned else 'pageable'} input): {nf} frames in {dt*1e3:.1f} ms = {nf/dt:.0f} frames/s ({nf*N/dt/1e9:.2f} GB/s of LLR input over PCIe)")
