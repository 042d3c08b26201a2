#!/usr/bin/env python3
"""tools/gen_crc8_golden.py -- BUILD CONTAINER ONLY. Known answers of the reference's CRC-8 remainder as bbdeheader_bb computes
it (gf2_poly_rem with the table of build_gf2_poly_rem_lut, lib/gf_util.h, through oracle/_ref/libdvbs2_ref_bch.so:
ref_crc8_rem) -> tests/golden/crc8_golden.json. Inputs: seeded random byte strings of the lengths the block uses (10, 188) and
others, plus strings whose check passes (remainder 0)."""
import json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fec_testlib as T

ref = T.ref_bch()
assert ref is not None, "run `make -C oracle` in the build container first"
rng = np.random.default_rng(20240928)
cases = []
for n in [1, 2, 3, 9, 10, 10, 10, 11, 17, 64, 187, 188, 188, 188, 189, 255, 256, 300, 1000]:
    d = rng.integers(0, 256, n, dtype=np.uint8)
    cases.append({"hex": d.tobytes().hex(), "rem": int(ref.ref_crc8_rem(T.ptr(d), n))})
for n in [10, 188, 188]:  # strings that end in their own CRC: remainder 0
    d = rng.integers(0, 256, n, dtype=np.uint8)
    d[-1] = T.crc8_dvbs2(d[:-1])
    r = int(ref.ref_crc8_rem(T.ptr(d), n))
    assert r == 0
    cases.append({"hex": d.tobytes().hex(), "rem": r})
json.dump({"generator": "0b111010101", "source": "reference lib/gf_util.h gf2_poly_rem via oracle/_ref (tools/gen_crc8_golden.py)", "cases": cases},
          open(os.path.join(ROOT, "tests", "golden", "crc8_golden.json"), "w"), indent=0)
print(len(cases), "cases")
