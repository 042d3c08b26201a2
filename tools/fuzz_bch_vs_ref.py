#!/usr/bin/env python3
"""Container only: fuzzes the plain-C restatement (oracle/bch_oracle.c) against the GENUINE reference BCH codec
(oracle/_ref/libdvbs2_ref_bch.so) -- message bytes and return codes, incl. the region beyond t errors (partial flips,
-1) and the two throw sites (crafted words). usage: fuzz_bch_vs_ref.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "gr-dvbs2rx_amd", "python"))
import numpy as np
import fec_testlib as T
import bch_craft as B
from dvbs2rx_amd import capi, get_fec_info
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 11)
combos = [(capi.FECFRAME_NORMAL, r) for r in ("C1_4", "C1_3", "C2_5", "C1_2", "C3_5", "C2_3", "C3_4", "C4_5", "C5_6", "C8_9", "C9_10", "C154_180")] + \
         [(capi.FECFRAME_SHORT, r) for r in ("C1_4", "C1_3", "C2_5", "C1_2", "C3_5", "C2_3", "C3_4", "C4_5", "C5_6", "C8_9")]
t0 = time.time(); total = 0; bad = 0; stats = {}
while time.time() - t0 < budget:
    fs, rate = combos[rng.integers(len(combos))]
    fi = get_fec_info(capi.STANDARD_DVBS2, fs, rate)
    n, k, t = fi["bch_n"], fi["bch_k"], fi["bch_t"]
    m, prim = T.BCH_FIELDS[fs]
    ref, ob, gf = T.RefBch(prim, t, n), T.OracleBch(m, prim, t, n), B.GF(m, prim)
    msg = rng.integers(0, 256, (24, k // 8), dtype=np.uint8)
    cw = ref.encode(msg)
    assert np.array_equal(cw, ob.encode_bytes(msg)), "encoders differ"
    rx = [T.flip_bits(cw[i], rng.choice(n, int(rng.integers(0, 4 * t + 1)), replace=False)) for i in range(20)]
    rx += [rng.integers(0, 256, n // 8, dtype=np.uint8) for _ in range(2)]
    rx += [B.word_from_exponents(n, B.craft_quadratic(gf, n, t, rng)), B.word_from_exponents(n, B.craft_beyond_n(gf, n, t, rng))]
    rx = np.stack(rx)
    want, wret = ref.decode(rx)
    got, gret = ob.decode_bytes(rx)
    for r in wret:
        stats[min(r, 1)] = stats.get(min(r, 1), 0) + 1
    if wret != gret.tolist() or not np.array_equal(want, got):
        bad += 1; print("MISMATCH", fs, rate, wret, gret.tolist(), flush=True)
    total += len(rx); ref.close()
print(f"oracle vs genuine reference BCH: {total} codewords (ret -2: {stats.get(-2, 0)}, -1: {stats.get(-1, 0)}, 0: {stats.get(0, 0)}, >0: {stats.get(1, 0)}), {bad} mismatching batches, {time.time() - t0:.0f} s")
sys.exit(1 if bad else 0)
