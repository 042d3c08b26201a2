#!/usr/bin/env python3
"""GPU box: randomized BCH parity soak: every (framesize, rate) the reference's bch_decoder_bb accepts, random error
counts 0..3t (incl. the failure region with partial corrections) vs the restatement. usage: fuzz_bch.py [seconds] [seed]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-dvbs2rx_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fec_testlib as T
from dvbs2rx_amd import BchDecoder, capi, get_fec_info
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
combos = [(capi.FECFRAME_NORMAL, r) for r in ("C1_4", "C1_3", "C2_5", "C1_2", "C3_5", "C2_3", "C3_4", "C4_5", "C5_6", "C8_9", "C9_10")] + \
         [(capi.FECFRAME_SHORT, r) for r in ("C1_4", "C1_3", "C2_5", "C1_2", "C3_5", "C2_3", "C3_4", "C4_5", "C5_6", "C8_9")]
t0 = time.time(); n = 0; fails = 0; throws = 0
while time.time() - t0 < budget:
    fs, rate = combos[rng.integers(len(combos))]
    fi = get_fec_info(capi.STANDARD_DVBS2, fs, rate)
    m, prim = T.BCH_FIELDS[fs]
    ob = T.OracleBch(m, prim, fi["bch_t"], fi["bch_n"])
    nf = 24
    msg = rng.integers(0, 256, (nf, ob.k // 8), dtype=np.uint8)
    cw = ob.encode_bytes(msg)
    rx = np.stack([T.flip_bits(cw[i], rng.choice(ob.n, int(rng.integers(0, 3 * ob.t + 1)), replace=False)) for i in range(nf)])
    dec = BchDecoder(framesize=fs, rate=rate, max_frames=nf)
    dec.set_descramble(bool(rng.integers(2)))
    out, ret = dec.work(rx)
    want, wret = ob.decode_bytes(rx)
    # compare without the descrambler: undo it through the oracle when it was on
    on = not np.array_equal(out, want) and np.array_equal(out, T.oracle_bb_descramble(want))
    ok = ret.tolist() == wret.tolist() and (np.array_equal(out, want) or on)
    throws += int((wret == -2).sum())
    dec.close(); n += nf
    if not ok:
        fails += 1; print("MISMATCH", fs, rate, flush=True)
print(f"bch fuzz: {n} codewords ({throws} in the reference-throws class), {fails} mismatching batches, {time.time() - t0:.0f} s")
sys.exit(1 if fails else 0)
