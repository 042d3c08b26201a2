#!/usr/bin/env python3
"""GPU box: randomized parity soak. Random table / amplitude / noise level / group size / iteration cap, GPU result
(LLRs, bits, group return values) against the genuine reference decoder (or the restatement), bit for bit.
usage: fuzz_ldpc.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-dvbs2rx_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fec_testlib as T
from dvbs2rx_amd import LdpcDecoder, capi, ldpc_table_names
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
names = ldpc_table_names()
t0 = time.time(); n = 0; fails = 0
while time.time() - t0 < budget:
    table = names[rng.integers(len(names))]
    N, K, _, _ = T.ldpc_info(table)
    G = int(rng.choice([32, 32, 16]))
    nf = G * int(rng.integers(1, 3))
    cap = int(rng.integers(2, 40))
    kind = rng.integers(4)
    if kind == 0:
        llr = T.llr_noise(nf, N, int(rng.integers(1 << 30)), sigma=float(rng.uniform(2, 60)))
    else:
        amp = float(rng.choice([3, 6, 12, 40, 100])); sigma = amp * float(rng.uniform(0.2, 1.4))
        llr, _ = T.llr_codeword_awgn(table, nf, int(rng.integers(1 << 30)), amp=amp, sigma=sigma)
        if kind == 3:
            llr[rng.integers(nf)] = 0  # an all-zero frame inside the group
    dec = LdpcDecoder(table=table, message_bits=K, group_size=G, max_frames=nf, max_trials=cap, outputmode=capi.OM_CODEWORD)
    bits, out, ret = dec.work(llr, want_llr=True); kn = dec.kernel_name; dec.close()
    if T.ref_ldpc() is not None:
        want, wret = T.ref_ldpc_decode(table, llr, 0 if G == 32 else 2, cap)
    else:
        want, wret = T.oracle_ldpc_decode(table, llr, G, cap)
    ok = ret.tolist() == wret and np.array_equal(out, want) and np.array_equal(bits, T.pack_bits(want, N))
    n += 1
    if not ok:
        fails += 1
        print("MISMATCH", table, kn, "G", G, "nf", nf, "cap", cap, "kind", kind, flush=True)
print(f"fuzz: {n} cases, {fails} mismatches, {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
