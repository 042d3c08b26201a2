#!/usr/bin/env python3
"""tools/gen_bch_kat.py -- BUILD CONTAINER ONLY. Transcribes the known-answer constants of the reference's own
BCH/GF/QPSK unit tests into tests/golden/bch_kat.json and tests/golden/demap_kat.json (DATA only: numbers that
the reference's tests assert, which are themselves taken from the DVB-S2 standard tables / textbook examples).

Sources parsed as text (Boost.UTF is not installed, the tests cannot be run here):
  lib/qa_gf.cc:204-284     minimal polynomials g1..g12 of GF(2^16), GF(2^14), GF(2^15) (EN 302 307 tables 6a/6b, S2X table 7)
  lib/qa_bch.cc:90-179     generator polynomials over GF(2^4) and GF(2^6)
  lib/qa_bch.cc:191-230    all 128 codewords of the (15,7) t=2 code
  lib/qa_bch.cc:276-281    syndrome of r(x) = x^8 + 1
  lib/qa_bch.cc:363-380    syndrome, sigma(x) and error-location numbers of r(x) = x^12 + x^5 + x^3 (t=3)
  lib/qa_qpsk.cc:67-79     soft demap known answer
"""
import json, re
REF = "/root/reference/lib/"
gf = open(REF + "qa_gf.cc").read()
bch = open(REF + "qa_bch.cc").read()

def bins(txt):
    return [int(x, 2) for x in re.findall(r"0b([01]+)", txt)]

# --- minimal polynomials
sec = gf[gf.index("test_gf2m_dvbs2_min_poly"):gf.index("test_gf2_poly_degrees")]
blocks = re.findall(r"expected_min_poly\d = \{(.*?)\};", sec, re.S)
prims = [0b10000000000101101, 0b100000000101011, 0b1000000000101101]
minpoly = [dict(m=m, prim_poly=p, polys=bins(b)) for m, p, b in zip([16, 14, 15], prims, blocks)]
assert all(len(x["polys"]) == 12 and x["polys"][0] == x["prim_poly"] for x in minpoly)

# --- generator polynomials GF(2^4)
sec = bch[bch.index("test_bch_gen_poly"):bch.index("test_bch_encoder")]
gen_m4 = [dict(t=1, g=0b10011), dict(t=2, g=0b111010001), dict(t=3, g=0b10100110111)]
for g in gen_m4:
    assert bin(g["g"])[2:] in sec
# GF(2^6): successive products; factors listed in the test
fac = [(1, 0b1000011), (2, 0b1010111), (3, 0b1100111), (4, 0b1001001), (5, 0b1101), (6, 0b1101101),
       (7, 0b1011011), (10, 0b1110101), (11, 0b111), (13, 0b1110011), (15, 0b1011)]
ks = {1: 57, 2: 51, 3: 45, 4: 39, 5: 36, 6: 30, 7: 24, 10: 18, 11: 16, 13: 10, 15: 7}
for t, f in fac:
    assert bin(f)[2:] in sec, (t, bin(f))
def pmul(a, b):
    r = 0
    while b:
        if b & 1: r ^= a
        a <<= 1; b >>= 1
    return r
gen_m6 = []; g = 1
for t, f in fac:
    g = pmul(g, f)
    gen_m6.append(dict(t=t, g=g, k=ks[t]))

# --- (15,7) codewords
sec = bch[bch.index("expected_codewords = {"):bch.index("T max_msg = (1 << codec.get_k()) - 1;")]
cws = bins(sec)
assert len(cws) == 128

kat = dict(minpoly=minpoly, gen_m4=gen_m4, gen_m6=gen_m6, codewords_15_7=cws,
           syndrome=dict(m=4, prim_poly=0b10011, t=2, rx=0b100000001, alpha_exps=[2, 4, 7, 8]),
           errloc=dict(m=4, prim_poly=0b10011, t=3, rx=0b1000000101000, syndrome_alpha_exps=[0, 0, 10, 0, 10, 5],
                       sigma_alpha_exps=[0, 0, None, 5], numbers_alpha_exps=[12, 5, 3]))
json.dump(kat, open("tests/golden/bch_kat.json", "w"))

q = open(REF + "qa_qpsk.cc").read()
sec = q[q.index("test_qpsk_soft_demap") if "test_qpsk_soft_demap" in q else 0:]
print(sec[:1200])
