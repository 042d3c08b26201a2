"""CPU-only: pins the plain-C oracle (oracle/) to the reference's own known-answer tests and golden digests, and
checks the host logic of the product library (parameter map, exported C symbols). No GPU compute here."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

import fec_testlib as T
from dvbs2rx_amd import capi, get_fec_info

GOLD = os.path.join(T.ROOT, "tests", "golden")
KAT = json.load(open(os.path.join(GOLD, "bch_kat.json")))


# ------------------------------------------------------------------ GF / BCH (lib/qa_gf.cc, lib/qa_bch.cc)
@pytest.mark.parametrize("entry", KAT["minpoly"], ids=lambda e: f"GF2^{e['m']}")
def test_dvbs2_minimal_polynomials(entry):
    b = T.OracleBch(entry["m"], entry["prim_poly"], 1)
    for i, want in enumerate(entry["polys"], start=1):
        assert b.minpoly(2 * i - 1) == want


def test_generator_polynomials():
    for e in KAT["gen_m4"]:
        assert T.OracleBch(4, 0b10011, e["t"]).genpoly_int() == e["g"]
    for e in KAT["gen_m6"]:
        b = T.OracleBch(6, 0b1000011, e["t"])
        assert b.genpoly_int() == e["g"] and b.k == e["k"]


def test_all_codewords_15_7():
    b = T.OracleBch(4, 0b10011, 2)
    assert (b.n, b.k) == (15, 7)
    for msg, want in enumerate(KAT["codewords_15_7"]):
        bits = np.array([(msg >> (6 - i)) & 1 for i in range(7)], np.uint8)
        cw = b.encode_bits(bits)
        assert int("".join(map(str, cw)), 2) == want


def test_syndrome_and_error_locator_kats():
    s = KAT["syndrome"]
    b = T.OracleBch(s["m"], s["prim_poly"], s["t"])
    rx = np.array([(s["rx"] >> (b.n - 1 - i)) & 1 for i in range(b.n)], np.uint8)
    assert b.syndrome_bits(rx).tolist() == [b.alpha(e) for e in s["alpha_exps"]]
    e = KAT["errloc"]
    b = T.OracleBch(e["m"], e["prim_poly"], e["t"])
    rx = np.array([(e["rx"] >> (b.n - 1 - i)) & 1 for i in range(b.n)], np.uint8)
    S = b.syndrome_bits(rx)
    assert S.tolist() == [b.alpha(x) for x in e["syndrome_alpha_exps"]]
    sigma, nums, cnt = b.err_loc(S)
    assert sigma.tolist() == [0 if x is None else b.alpha(x) for x in e["sigma_alpha_exps"]]
    assert cnt == 3 and nums.tolist() == [b.alpha(x) for x in e["numbers_alpha_exps"]]


def test_exhaustive_one_and_two_bit_errors_32_8():
    """lib/qa_bch.cc:539-604: shortened (32, 8) code over GF(2^6), t = 4."""
    b = T.OracleBch(6, 0b1000011, 4, 32)
    assert (b.n, b.k) == (32, 8)
    msg = np.array([[0xA7]], np.uint8)
    cw = b.encode_bytes(msg)[0]
    for i in range(32):
        for j in range(i, 32):
            pos = [i] if i == j else [i, j]
            out, ret = b.decode_bytes(T.flip_bits(cw, pos)[None])
            assert ret[0] == len(pos) and out[0, 0] == 0xA7


def test_all_dvbs2_codes_correct_t_errors():
    """lib/qa_bch.cc:652-747: every (n, t) of the parameter table: encode, t random errors, decode."""
    rows = json.load(open(os.path.join(GOLD, "fec_params.json")))["rows"]
    seen = set()
    rng = np.random.default_rng(11)
    for r in rows:
        key = (r["framesize_id"], r["bch_n"], r["bch_t"])
        if key in seen or r["bch_n"] % 8 or r["bch_k"] % 8:
            continue
        seen.add(key)
        m, prim = T.BCH_FIELDS[r["framesize_id"]]
        b = T.OracleBch(m, prim, r["bch_t"], r["bch_n"])
        assert b.k == r["bch_k"], r
        msg = rng.integers(0, 256, (1, b.k // 8), dtype=np.uint8)
        cw = b.encode_bytes(msg)
        pos = rng.choice(b.n, r["bch_t"], replace=False)
        out, ret = b.decode_bytes(T.flip_bits(cw[0], pos)[None])
        assert ret[0] == r["bch_t"] and np.array_equal(out, msg)
    assert len(seen) >= 40


def test_beyond_t_returns_failure():
    """lib/qa_bch.cc:606-650: more than t errors -> -1 (or the exception path, reported as -2)."""
    b = T.OracleBch(14, 0b100000000101011, 12, 3240)
    rng = np.random.default_rng(5)
    msg = rng.integers(0, 256, (1, b.k // 8), dtype=np.uint8)
    cw = b.encode_bytes(msg)[0]
    rets = []
    for trial in range(30):
        pos = rng.choice(b.n, 13 + trial, replace=False)
        rets.append(int(b.decode_bytes(T.flip_bits(cw, pos)[None])[1][0]))
    assert all(r in (-1, -2) for r in rets)


# ------------------------------------------------------------------ demapper (lib/qa_qpsk.cc:67-79)
def test_qpsk_soft_demap_kat():
    k = json.load(open(os.path.join(GOLD, "demap_kat.json")))
    syms = np.array([complex(a, b) for a, b in k["syms"]], np.complex64)[None]
    out = T.oracle_demap(syms, np.float32(2 * np.sqrt(2)), 4)
    assert out[0].tolist() == k["expected"]


# ------------------------------------------------------------------ LDPC golden digests (from the genuine reference)
LG = json.load(open(os.path.join(GOLD, "ldpc_golden.json")))


@pytest.mark.parametrize("case", LG, ids=lambda c: f"{c['table']}-{c['kind']}-{c['trials']}")
def test_ldpc_oracle_matches_reference_digests(case):
    x = T.make_input(case["table"], case["kind"], case["n_frames"], **case["params"])
    assert T.sha(x) == case["input_sha256"], "input generator drifted"
    N = x.shape[1]
    for G in ("32", "16"):
        y, ret = T.oracle_ldpc_decode(case["table"], x, int(G), case["trials"])
        want = case["results"][G]
        assert ret == want["ret"]
        assert T.sha(y) == want["llr_sha256"]
        assert T.sha(T.pack_bits(y, N)) == want["bits_sha256"]


def test_ldpc_oracle_vs_reference_live():
    """Only where oracle/_ref exists (it is built in the container that has /root/reference and travels as a .so)."""
    if T.ref_ldpc() is None:
        pytest.skip("oracle/_ref not built")
    x = T.make_input("S2_TABLE_C3", "awgn", 32, seed=21, amp=5, sigma=5.5)
    a, ra = T.oracle_ldpc_decode("S2_TABLE_C3", x, 32, 20)
    b, rb = T.ref_ldpc_decode("S2_TABLE_C3", x, 0, 20)
    assert ra == rb and np.array_equal(a, b)


# ------------------------------------------------------------------ product host logic (no device needed)
def test_bb_descrambler_sequence_known_answer():
    """DVB energy-dispersal PRBS (EN 302 307-1 clause 5.2.2 / EN 300 421 clause 4.4.1): published first bytes; the
    product's host-side sequence equals the oracle's over a whole normal BBFRAME."""
    from dvbs2rx_amd import bb_descramble_sequence
    seq = np.zeros(8100, np.uint8)
    T.oracle().oracle_bb_sequence(T.ptr(seq), 8100)
    assert bytes(seq[:8]).hex() == "03f6083430b8a393"
    assert np.array_equal(bb_descramble_sequence(8100), seq)
    x = np.arange(2 * 4026, dtype=np.uint32).astype(np.uint8).reshape(2, 4026)
    assert np.array_equal(T.oracle_bb_descramble(T.oracle_bb_descramble(x)), x)  # involution
    assert np.array_equal(T.oracle_bb_descramble(x)[1], x[1] ^ seq[:4026])


@pytest.mark.parametrize("gold", [0, 1, 131071, 262141])
def test_pl_scrambling_sequence_definition_vs_register_masks(gold):
    """Rn of ETSI EN 302 307-1 clause 5.5.4 two ways: the product evaluates the definition (x and y m-sequences,
    z_n(i) = x(i + n) + y(i), Rn = 2 z_n(i + 131072) + z_n(i)), the oracle follows the reference's shift registers with
    the 0x8050 / 0xFF60 masks (lib/pl_descrambler.cc:62-98). Whole maximum payload."""
    from dvbs2rx_amd import pl_scrambling_rn
    n = 360 * 90 + 22 * 36
    rn = np.zeros(n, np.uint8)
    T.oracle().oracle_pl_rn(gold, T.ptr(rn), n)
    assert rn.max() == 3 and np.array_equal(pl_scrambling_rn(gold, n), rn)
    if gold == 0:  # x(0) = 1, y(0) = 1 => z(0) = 0; the sequence is balanced
        assert rn[0] & 1 == 0 and abs(np.bincount(rn, minlength=4) / n - 0.25).max() < 0.02


def test_fec_params_match_reference():
    """gr-dvbs2rx_amd's parameter map vs the reference's get_fec_info() + table switch (tests/golden/fec_params.json)."""
    g = json.load(open(os.path.join(GOLD, "fec_params.json")))
    assert capi.lib.dvbs2_rate_name(3) == b"C1_2" and capi.lib.dvbs2_rate_from_name(b"C9_10") == g["rates"].index("C9_10")
    for r in g["rows"]:
        fi = get_fec_info(r["standard_id"], r["framesize_id"], r["rate_id"])
        assert (fi["bch_k"], fi["bch_n"], fi["bch_t"], fi["ldpc_k"], fi["ldpc_n"], fi["table"]) == \
               (r["bch_k"], r["bch_n"], r["bch_t"], r["ldpc_k"], r["ldpc_n"], r["table"]), r
    have = {(r["standard_id"], r["framesize_id"], r["rate_id"]) for r in g["rows"]}
    fi = capi.FecInfo()
    for s in range(2):
        for f in range(3):
            for rate in range(len(g["rates"])):
                rc = capi.lib.dvbs2_get_fec_info(s, f, rate, fi)
                assert (rc == 0) == ((s, f, rate) in have)


def test_c_abi_exports_every_declared_symbol():
    hdr = open(os.path.join(T.ROOT, "include", "dvbs2_fec_hip.h")).read()
    declared = set(re.findall(r"\b(dvbs2_[a-z0-9_]+)\s*\(", hdr))
    declared -= {n for n in declared if n.endswith("_t")}
    assert declared == set(capi.SYMBOLS), declared ^ set(capi.SYMBOLS)
    for name in declared:
        assert getattr(capi.lib, name) is not None


def test_create_fails_loudly_without_device_or_with_bad_args():
    h = C.c_void_p()
    if capi.lib.dvbs2_device_count() == 0:
        assert capi.lib.dvbs2_ldpc_create(C.byref(h), 0, 1, 3, 32, 8, 0) == capi.EDEVICE
        assert b"no HIP device" in capi.lib.dvbs2_last_error()
        assert capi.lib.dvbs2_bch_create(C.byref(h), 0, 1, 3, 8, 0) == capi.EDEVICE
        assert capi.lib.dvbs2_demap_create(C.byref(h), 1, 3, 0, 8, 0) == capi.EDEVICE
    assert capi.lib.dvbs2_ldpc_create(C.byref(h), 0, 1, 50, 32, 8, 0) == capi.EINVAL  # C_OTHER
    assert capi.lib.dvbs2_ldpc_create_table(C.byref(h), b"NOPE", 8, 1, 1, 0) == capi.EINVAL


def test_enums_match_reference():
    """The C ABI, the ctypes binding and the C++ host mirror take the reference's enumerator values
    (include/gnuradio/dvbs2rx/dvb_config.h, transcribed as data by tools/gen_enum_golden.py)."""
    import json, re
    gold = json.load(open(os.path.join(T.ROOT, "tests", "golden", "dvb_config_enums.json")))["enums"]
    hdr = open(os.path.join(T.ROOT, "include", "dvbs2_fec_hip.h")).read()
    macros = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+DVBS2_(\w+)\s+(-?\d+)\s", hdr)}
    flat = {}
    for e in gold.values():
        flat.update(e)
    checked = 0
    for name in ("STANDARD_DVBS2", "STANDARD_DVBT2", "FECFRAME_SHORT", "FECFRAME_NORMAL", "FECFRAME_MEDIUM",
                 "OM_CODEWORD", "OM_MESSAGE", "MOD_QPSK", "MOD_8PSK"):
        assert macros[name] == flat[name], name
        assert getattr(capi, name) == flat[name], name
        checked += 1
    assert checked == 9
    # the C++ host mirror spells the enumerations out: parse them the same way
    host = open(os.path.join(T.ROOT, "gr-dvbs2rx_amd", "host", "dvbs2rx_hip_blocks.h")).read()
    for m in re.finditer(r"enum\s+(\w+)\s*\{([^}]*)\}", host):
        val = -1
        for item in [x.strip() for x in m.group(2).split(",") if x.strip()]:
            if "=" in item:
                k, v = [y.strip() for y in item.split("=")]
                val = int(v, 0)
            else:
                k, val = item, val + 1
            assert gold[m.group(1)][k] == val, (m.group(1), k)
    # rate enumerators: the name table of the library follows dvb_code_rate_t
    for k, v in gold["dvb_code_rate_t"].items():
        if k != "C_OTHER":
            assert capi.lib.dvbs2_rate_from_name(k.encode()) == v, k


BCH_GOLD = json.load(open(os.path.join(GOLD, "bch_golden.json")))


@pytest.mark.parametrize("code", BCH_GOLD["codes"], ids=lambda c: f"n{c['n']}_t{c['t']}")
def test_bch_oracle_vs_reference_digests(code):
    """oracle/bch_oracle.c against digests taken from the GENUINE reference codec (tools/gen_bch_golden.py): every
    BASELINE (n, t): 0..t errors, beyond t (-1, partial flips), parity-only, garbage, and both throw sites (-2)."""
    m, prim = T.BCH_FIELDS[code["framesize"]]
    ob = T.OracleBch(m, prim, code["t"], code["n"])
    assert (ob.n, ob.k) == (code["n"], code["k"])
    seen = set()
    for c in code["cases"]:
        rx = T.bch_golden_input(ob, code["n"], code["k"], c)
        assert T.sha(rx) == c["sha_in"], c["name"]
        msg, ret = ob.decode_bytes(rx[None])
        assert int(ret[0]) == c["ret"], c["name"]
        assert T.sha(msg[0]) == c["sha_out"], c["name"]
        seen.add(c["ret"])
    assert {-2, -1, 0, 1, 2, 3, code["t"]} <= seen


def test_bch_oracle_vs_reference_live():
    """When oracle/_ref/libdvbs2_ref_bch.so is present (it is built in the container and travels to the GPU box): random
    words incl. the failure region, restatement vs the genuine codec."""
    if T.ref_bch() is None:
        pytest.skip("oracle/_ref/libdvbs2_ref_bch.so absent (reference sources not on this box and no prebuilt copy)")
    rng = np.random.default_rng(99)
    for fs, n, t in ((0, 3240, 12), (1, 32400, 12), (1, 58320, 8)):
        m, prim = T.BCH_FIELDS[fs]
        ref, ob = T.RefBch(prim, t, n), T.OracleBch(m, prim, t, n)
        msg = rng.integers(0, 256, (12, ob.k // 8), dtype=np.uint8)
        cw = ref.encode(msg)
        assert np.array_equal(cw, ob.encode_bytes(msg))
        rx = np.stack([T.flip_bits(cw[i], rng.choice(n, [0, 1, 2, t, t + 1, 15, 20, 30, 40, 41, 42, 100][i], replace=False)) for i in range(12)])
        want, wret = ref.decode(rx)
        got, gret = ob.decode_bytes(rx)
        assert wret == gret.tolist() and np.array_equal(want, got)
        ref.close()
