"""GPU parity: BCH decoder, soft demapper and the fused chain through the C ABI vs the CPU oracle, bit-exact."""
import json
import os

import numpy as np
import pytest

import fec_testlib as T
from dvbs2rx_amd import BchDecoder, Demapper, FecChain, LdpcDecoder, PlPayload, bb_descramble_sequence, capi, get_fec_info

pytestmark = pytest.mark.gpu
GOLD = os.path.join(T.ROOT, "tests", "golden")


def bch_pair(framesize, rate):
    fi = get_fec_info(capi.STANDARD_DVBS2, framesize, rate)
    m, prim = T.BCH_FIELDS[framesize]
    return T.OracleBch(m, prim, fi["bch_t"], fi["bch_n"]), fi


@pytest.mark.parametrize("framesize,rate", [(capi.FECFRAME_NORMAL, "C1_2"), (capi.FECFRAME_NORMAL, "C3_4"),
                                            (capi.FECFRAME_NORMAL, "C9_10"), (capi.FECFRAME_SHORT, "C1_4"),
                                            (capi.FECFRAME_NORMAL, "C2_3"), (capi.FECFRAME_SHORT, "C8_9")])
@pytest.mark.parametrize("syndromes", ["table", "product"])
def test_bch_error_patterns(framesize, rate, syndromes, monkeypatch):
    # both syndrome stages on the same words: one table read per set bit (small batches), the batched GF(2) matrix product (>= 32 frames)
    monkeypatch.setenv("DVBS2_BCH_SYND_MIN", "1" if syndromes == "product" else "1000000")
    ob, fi = bch_pair(framesize, rate)
    t, n, k = fi["bch_t"], fi["bch_n"], fi["bch_k"]
    rng = np.random.default_rng(17)
    # 0, 1, 2, 3, t, t+1, 40 errors anywhere; errors in the parity part only; two far apart / adjacent
    counts = [0, 1, 1, 1, 2, 2, 2, 2, 3, 3, t - 1, t, t, t + 1, t + 1, t + 2, 20, 40, 40, 100]
    msg = rng.integers(0, 256, (len(counts) + 4, k // 8), dtype=np.uint8)
    cw = ob.encode_bytes(msg)
    rx = []
    for i, c in enumerate(counts):
        rx.append(T.flip_bits(cw[i], rng.choice(n, c, replace=False)))
    base = len(counts)
    rx.append(T.flip_bits(cw[base], k + rng.choice(n - k, 3, replace=False)))        # parity bits only
    rx.append(T.flip_bits(cw[base + 1], [0, n - 1]))                                  # first and last bit
    rx.append(T.flip_bits(cw[base + 2], [k - 1, k]))                                  # message/parity boundary
    rx.append(T.flip_bits(cw[base + 3], list(range(100, 100 + t))))                   # burst of t
    rx = np.stack(rx)
    dec = BchDecoder(framesize=framesize, rate=rate, max_frames=len(rx))
    assert (dec.n, dec.k, dec.t) == (n, k, t)
    assert sum(int(b) << i for i, b in enumerate(dec.genpoly())) == ob.genpoly_int()
    out, ret = dec.work(rx)
    want, wret = ob.decode_bytes(rx)
    assert ret.tolist() == wret.tolist()
    assert np.array_equal(out, want)
    ok = [i for i, c in enumerate(counts) if c <= t]
    assert np.array_equal(out[ok], msg[ok]) and ret[ok].tolist() == [counts[i] for i in ok]
    assert dec.frame_error_cnt == int((wret == -1).sum())
    dec.close()


@pytest.mark.parametrize("framesize,rate,nf", [(capi.FECFRAME_NORMAL, "C3_4", 77), (capi.FECFRAME_SHORT, "C1_4", 133),
                                               (capi.FECFRAME_NORMAL, "C9_10", 33), (capi.FECFRAME_NORMAL, "C3_5", 64)])
def test_bch_batched_syndromes_ragged_batches(framesize, rate, nf):
    """The batched syndrome product (32 frames per wave, columns cut into chunks, partial last tile) on batches that are no multiple of 32:
    0 .. t + 2 errors per word, every frame against the oracle; a second call on the same handle with fewer frames (stale syndrome words
    of the first call must not leak)."""
    ob, fi = bch_pair(framesize, rate)
    t, n, k = fi["bch_t"], fi["bch_n"], fi["bch_k"]
    rng = np.random.default_rng(nf)
    msg = rng.integers(0, 256, (nf, k // 8), dtype=np.uint8)
    cw = ob.encode_bytes(msg)
    rx = np.stack([T.flip_bits(cw[i], rng.choice(n, int(rng.integers(0, t + 3)), replace=False)) for i in range(nf)])
    dec = BchDecoder(framesize=framesize, rate=rate, max_frames=nf)
    for m in (nf, 40 if nf > 40 else 32):
        out, ret = dec.work(rx[:m])
        want, wret = ob.decode_bytes(rx[:m])
        assert ret.tolist() == wret.tolist()
        assert np.array_equal(out, want)
    dec.close()


@pytest.mark.parametrize("framesize,rate", [(capi.FECFRAME_NORMAL, "C1_2"), (capi.FECFRAME_SHORT, "C1_4"),
                                            (capi.FECFRAME_NORMAL, "C9_10")])
def test_bch_fused_bb_descrambler(framesize, rate):
    """bch_decoder_bb -> bbdescrambler_bb (reference lib/bbdescrambler_bb_impl.cc:67-82) in one kernel: corrected,
    uncorrectable and clean frames; switching the fusion off restores the plain output."""
    ob, fi = bch_pair(framesize, rate)
    t, n, k = fi["bch_t"], fi["bch_n"], fi["bch_k"]
    dec = BchDecoder(framesize=framesize, rate=rate, max_frames=6)
    rng = np.random.default_rng(77)
    msg = rng.integers(0, 256, (6, k // 8), dtype=np.uint8)
    cw = ob.encode_bytes(msg)
    rx = np.stack([T.flip_bits(cw[i], rng.choice(n, [0, 1, t, t + 3, 40, 2][i], replace=False)) for i in range(6)])
    want, wret = ob.decode_bytes(rx)
    plain, ret = dec.work(rx)
    assert np.array_equal(plain, want) and ret.tolist() == wret.tolist()
    dec.set_descramble(True)
    got, ret = dec.work(rx)
    assert ret.tolist() == wret.tolist()
    assert np.array_equal(got, T.oracle_bb_descramble(want))
    assert np.array_equal(got[0] ^ msg[0], bb_descramble_sequence(k // 8))
    dec.set_descramble(False)
    assert np.array_equal(dec.work(rx)[0], want)
    dec.close()


def test_bch_medium_frames_rejected_like_the_reference():
    """GF(2^15), t = 12 gives 180 parity bits: k = n - 180 is not a multiple of 8, so the reference's byte API
    throws "u8 array messages are only supported for n and k multiple of 8." (lib/bch.cc:19-24; qa_bch.cc:743
    excludes medium frames). The shim reports the same text through DVBS2_EINVAL."""
    import ctypes as C
    h = C.c_void_p()
    rc = capi.lib.dvbs2_bch_create(C.byref(h), capi.STANDARD_DVBS2, capi.FECFRAME_MEDIUM,
                                   capi.lib.dvbs2_rate_from_name(b"C1_3_MEDIUM"), 4, 0)
    assert rc == capi.EINVAL and b"multiple of 8" in capi.lib.dvbs2_last_error()


def test_bch_failure_region_matches_oracle():
    """> t errors: partial corrections, -1, and the would-throw cases (-2) must match the restated reference."""
    ob, fi = bch_pair(capi.FECFRAME_SHORT, "C1_4")
    rng = np.random.default_rng(23)
    msg = rng.integers(0, 256, (256, fi["bch_k"] // 8), dtype=np.uint8)
    cw = ob.encode_bytes(msg)
    rx = np.stack([T.flip_bits(cw[i], rng.choice(fi["bch_n"], 13 + i % 30, replace=False)) for i in range(256)])
    # plus random garbage (what a failed LDPC decode hands over)
    rx = np.concatenate([rx, rng.integers(0, 256, (64, fi["bch_n"] // 8), dtype=np.uint8)])
    dec = BchDecoder(framesize=capi.FECFRAME_SHORT, rate="C1_4", max_frames=len(rx))
    out, ret = dec.work(rx)
    want, wret = ob.decode_bytes(rx)
    assert ret.tolist() == wret.tolist()
    assert np.array_equal(out, want)
    assert set(np.unique(wret)) <= {-1, -2}
    dec.close()


def test_bch_reference_digests_on_gpu():
    """The kernel against digests of the GENUINE reference codec (tests/golden/bch_golden.json <- tools/gen_bch_golden.py):
    every BASELINE (n, t) x {0, 1, 2, 3, t, t+1, 40, 100 errors, parity only, garbage} plus words crafted to reach BOTH
    places where the reference throws: a degree-2 locator without roots (lib/bch.cc:359-367 -> lib/gf.h:110) and an
    error location >= n on the shortened code (lib/bch.cc:443-444), reported as -2."""
    import json, os
    gold = json.load(open(os.path.join(T.ROOT, "tests", "golden", "bch_golden.json")))
    for code in gold["codes"]:
        m, prim = T.BCH_FIELDS[code["framesize"]]
        ob = T.OracleBch(m, prim, code["t"], code["n"])
        rx = np.stack([T.bch_golden_input(ob, code["n"], code["k"], c) for c in code["cases"]])
        dec = BchDecoder(framesize=code["framesize"], rate=code["rate"], max_frames=len(rx))
        out, ret = dec.work(rx)
        assert ret.tolist() == [c["ret"] for c in code["cases"]], code["rate"]
        for i, c in enumerate(code["cases"]):
            assert T.sha(rx[i]) == c["sha_in"] and T.sha(out[i]) == c["sha_out"], (code["rate"], c["name"])
        assert ret.tolist().count(-2) == 2
        dec.close()


def test_bch_vs_genuine_reference_live():
    """Random words incl. > t errors and garbage against the genuine codec when oracle/_ref holds it."""
    if T.ref_bch() is None:
        pytest.skip("oracle/_ref/libdvbs2_ref_bch.so absent")
    rng = np.random.default_rng(5)
    for fs, rate in ((capi.FECFRAME_SHORT, "C1_4"), (capi.FECFRAME_NORMAL, "C1_2"), (capi.FECFRAME_NORMAL, "C9_10")):
        fi = get_fec_info(capi.STANDARD_DVBS2, fs, rate)
        n, k, t = fi["bch_n"], fi["bch_k"], fi["bch_t"]
        ref = T.RefBch(T.BCH_FIELDS[fs][1], t, n)
        msg = rng.integers(0, 256, (48, k // 8), dtype=np.uint8)
        cw = ref.encode(msg)
        rx = np.stack([T.flip_bits(cw[i], rng.choice(n, int(rng.integers(0, 4 * t)), replace=False)) for i in range(40)] +
                      [rng.integers(0, 256, n // 8, dtype=np.uint8) for _ in range(8)])
        want, wret = ref.decode(rx)
        dec = BchDecoder(framesize=fs, rate=rate, max_frames=len(rx))
        out, ret = dec.work(rx)
        assert ret.tolist() == wret and np.array_equal(out, want)
        dec.close(); ref.close()


def test_bch_small_field_kats_on_gpu():
    """(32, 8) t=4 shortened code over GF(2^6) (lib/qa_bch.cc:539-604), all 1- and 2-bit patterns."""
    ob = T.OracleBch(6, 0b1000011, 4, 32)
    cw = ob.encode_bytes(np.array([[0xA7]], np.uint8))[0]
    pats = [[i] for i in range(32)] + [[i, j] for i in range(32) for j in range(i + 1, 32)]
    rx = np.stack([T.flip_bits(cw, p) for p in pats])
    dec = BchDecoder(raw=(6, 0b1000011, 4, 32), max_frames=len(rx))
    out, ret = dec.work(rx)
    assert ret.tolist() == [len(p) for p in pats] and (out == 0xA7).all()
    dec.close()


# ------------------------------------------------------------------ demapper
def rand_syms(nf, ns, seed, sigma):
    rng = np.random.default_rng(seed)
    ph = rng.integers(0, 8, (nf, ns)) * (np.pi / 4) + np.pi / 4
    return (np.exp(1j * ph) + sigma * (rng.normal(size=(nf, ns)) + 1j * rng.normal(size=(nf, ns)))).astype(np.complex64)


def test_qpsk_demap_bit_exact_and_kat():
    dm = Demapper(framesize=capi.FECFRAME_NORMAL, rate="C1_2", constellation=capi.MOD_QPSK, max_frames=6)
    assert (dm.n_syms, dm.n_llr, dm.n_mod) == (32400, 64800, 2)
    syms = rand_syms(6, 32400, 1, 0.5)
    syms[0, :4] = [1 + 1j, 1 - 1j, -1 - 1j, -1 + 1j]
    n0 = np.array([2 * np.sqrt(2), 0.05, 0.3, 1.0, 0.011, 7.7], np.float32)
    out = dm.work(syms, n0)
    assert out[0, :8].tolist() == json.load(open(os.path.join(GOLD, "demap_kat.json")))["expected"]
    assert np.array_equal(out, T.oracle_demap(syms, n0, 4))
    assert np.array_equal(dm.work(syms, 0.3), T.oracle_demap(syms, np.float32(0.3), 4))  # scalar N0
    snr = dm.estimate_snr(syms)
    want = [T.oracle().oracle_demap_snr(T.ptr(np.ascontiguousarray(syms[f])), 32400, 4) for f in range(6)]
    assert np.allclose(snr, want, rtol=2e-4)  # float reduction order differs (tolerance stated in DESIGN.md)
    dm.close()


@pytest.mark.parametrize("rate,order", [("C3_4", 0), ("C3_5", 1), ("C25_36", 2), ("C8_15", 2)])
def test_8psk_demap_bit_exact(rate, order):
    fs = capi.FECFRAME_SHORT if rate == "C8_15" else capi.FECFRAME_NORMAL
    dm = Demapper(framesize=fs, rate=rate, constellation=capi.MOD_8PSK, max_frames=3)
    assert dm.column_order == order and dm.n_mod == 3
    syms = rand_syms(3, dm.n_syms, 2, 0.2)
    n0 = np.array([0.05, 0.3, 1.0], np.float32)
    out = dm.work(syms, n0)
    assert np.array_equal(out, T.oracle_demap(syms, n0, 8, order))
    snr = dm.estimate_snr(syms)
    want = [T.oracle().oracle_demap_snr(T.ptr(np.ascontiguousarray(syms[f])), dm.n_syms, 8) for f in range(3)]
    assert np.allclose(snr, want, rtol=2e-4)
    dm.close()


@pytest.mark.parametrize("constellation,rate,order", [(capi.MOD_QPSK, "C1_2", 0), (capi.MOD_8PSK, "C3_4", 0),
                                                      (capi.MOD_8PSK, "C3_5", 1), (capi.MOD_8PSK, "C25_36", 2)])
def test_snr_refinement_from_decoded_llrs(constellation, rate, order):
    """handle_llr_pdu's per-frame estimate (reference :268-307): reference points from the signs of decoded LLRs."""
    dm = Demapper(framesize=capi.FECFRAME_NORMAL, rate=rate, constellation=constellation, max_frames=4)
    rng = np.random.default_rng(11)
    llr = rng.integers(-128, 128, (4, dm.n_llr)).astype(np.int8)
    llr[0, :64] = 0  # zero LLR counts as positive
    # symbols = points re-mapped from those LLRs + noise of known power => snr ~ 1 / (2 sigma^2)
    bits = (llr < 0).astype(np.uint8)
    if constellation == capi.MOD_QPSK:
        pts = ((1 - 2.0 * bits[:, 0::2]) + 1j * (1 - 2.0 * bits[:, 1::2])) * np.sqrt(0.5)
    else:
        rows = dm.n_syms
        ra = {0: (0, rows, 2 * rows), 1: (2 * rows, rows, 0), 2: (rows, 0, 2 * rows)}[order]
        pts = T.map_8psk(np.stack([bits[:, ra[c]:ra[c] + rows] for c in range(3)], axis=-1))
    sigma = np.array([0.05, 0.2, 0.4, 0.7])[:, None]
    syms = (pts + sigma * (rng.normal(size=pts.shape) + 1j * rng.normal(size=pts.shape))).astype(np.complex64)
    snr = dm.refine_snr(syms, llr)
    want = [T.oracle().oracle_demap_snr_refined(T.ptr(np.ascontiguousarray(syms[f])), T.ptr(llr[f]), dm.n_syms,
                                               4 if constellation == capi.MOD_QPSK else 8, order) for f in range(4)]
    assert np.allclose(snr, want, rtol=2e-4)  # float reduction order differs from the sequential restatement
    assert np.allclose(snr, 1 / (2 * sigma[:, 0] ** 2), rtol=0.05)
    dm.close()


def test_unsupported_constellation_rejected():
    import ctypes as C
    h = C.c_void_p()
    assert capi.lib.dvbs2_demap_create(C.byref(h), 1, 3, 3, 4, 0) == capi.EINVAL  # MOD_16APSK
    assert b"Unsupported constellation" in capi.lib.dvbs2_last_error()


# ------------------------------------------------------------------ PLFRAME payload step
@pytest.mark.parametrize("n_slots,has_pilots,gold", [(360, True, 0), (360, False, 0), (240, True, 5), (90, True, 131071),
                                                     (144, False, 7), (36, True, 0)])
def test_plframe_payload_step(n_slots, has_pilots, gold):
    """Descramble + pilot removal + per-segment de-rotation vs the restatement of plsync_cc_impl::handle_payload():
    coarse-corrected frames (rotator restarted from the pilot phases), not coarse-corrected ones (PLHEADER phase only),
    and the inverse property: scrambling + rotating known symbols and running the step gives them back."""
    pp = PlPayload(gold_code=gold, n_slots=n_slots, has_pilots=has_pilots, max_frames=4)
    npil = ((n_slots - 1) >> 4) if has_pilots else 0
    assert (pp.payload_len, pp.xfecframe_len, pp.n_pilots) == (90 * n_slots + 36 * npil, 90 * n_slots, npil)
    rng = np.random.default_rng(n_slots + gold)
    nf = 4
    payload = (rng.normal(size=(nf, pp.payload_len)) + 1j * rng.normal(size=(nf, pp.payload_len))).astype(np.complex64) * 0.7
    hph = rng.uniform(-np.pi, np.pi, nf).astype(np.float32)
    fine = np.array([2.5e-4, -1.0e-4, 3.3e-4, 0.0], np.float32)
    coarse = np.array([1, 1, 0, 1], np.int32)
    pil = rng.uniform(-np.pi, np.pi, (nf, max(npil, 1))).astype(np.float32)
    out = pp.work(payload, hph, fine, coarse, pil)
    want = T.oracle_pl_payload(payload, n_slots, has_pilots, gold, hph, fine, coarse, pil)
    assert np.abs(out - want).max() < 1e-4 * 4  # float rotator recurrence vs direct phase; |symbols| up to ~4
    # inverse property on frame 0: x -> scramble(x) * exp(+j phase) -> step -> x
    from dvbs2rx_amd import pl_scrambling_rn
    rn = pl_scrambling_rn(gold, pp.payload_len)
    x = (rng.normal(size=pp.xfecframe_len) + 1j * rng.normal(size=pp.xfecframe_len)).astype(np.complex64)
    tx = np.zeros(pp.payload_len, np.complex128)
    o = np.arange(pp.xfecframe_len)
    blk = (o // 90 // 16) if has_pilots else np.zeros_like(o)
    k = o + 36 * blk
    inc = 2 * np.pi * float(fine[0])
    theta0 = np.where(blk > 0, pil[0][np.maximum(blk - 1, 0)], hph[0]).astype(np.float64)
    steps = np.where(blk > 0, o - blk * 1440, o)
    tx[k] = x * np.exp(1j * (np.pi / 2) * rn[k]) * np.exp(1j * (theta0 + inc * steps))
    back = pp.work(np.broadcast_to(tx.astype(np.complex64), (1, pp.payload_len)), hph[:1], fine[:1], coarse[:1], pil[:1])
    assert np.abs(back[0] - x).max() < 2e-5 * 8
    pp.close()


def test_plframe_payload_rejects_bad_arguments():
    import ctypes as C
    h = C.c_void_p()
    assert capi.lib.dvbs2_plpayload_create(C.byref(h), 0, 400, 1, 4, 0) == capi.EINVAL   # more than MAX_SLOTS
    assert capi.lib.dvbs2_plpayload_create(C.byref(h), 1 << 18, 360, 1, 4, 0) == capi.EINVAL


# ------------------------------------------------------------------ chain (BASELINE config 3 shape, small batch)
def test_chain_8psk_3_4_normal():
    import torch
    fi = get_fec_info(capi.STANDARD_DVBS2, capi.FECFRAME_NORMAL, "C3_4")
    ob, _ = bch_pair(capi.FECFRAME_NORMAL, "C3_4")
    nf, G = 32, 32
    rng = np.random.default_rng(31)
    msg = rng.integers(0, 256, (nf, fi["bch_k"] // 8), dtype=np.uint8)
    bch_cw = ob.encode_bytes(msg)
    cw = T.ldpc_encode(fi["table"], np.unpackbits(bch_cw, axis=1))
    rows = 21600
    syms = T.map_8psk(np.stack([cw[:, :rows], cw[:, rows:2 * rows], cw[:, 2 * rows:]], axis=-1))
    es_n0_db = 8.5
    n0 = np.float32(10 ** (-es_n0_db / 10))
    noise = np.sqrt(n0 / 2) * (rng.normal(size=syms.shape) + 1j * rng.normal(size=syms.shape))
    rx = (syms + noise).astype(np.complex64)
    rx[nf - 1] = (noise[nf - 1] * 3).astype(np.complex64)  # one hopeless frame: LDPC fails, BCH sees garbage
    chain = FecChain(rate="C3_4", constellation=capi.MOD_8PSK, group_size=G, max_frames=nf, max_trials=30)
    assert (chain.n_syms, chain.msg_bytes) == (rows, fi["bch_k"] // 8)
    d_syms = torch.from_numpy(rx.view(np.float32).reshape(nf, -1)).cuda()
    d_n0 = torch.tensor([n0], dtype=torch.float32, device="cuda")
    d_msg = torch.empty((nf, chain.msg_bytes), dtype=torch.uint8, device="cuda")
    d_ret = torch.empty(1, dtype=torch.int32, device="cuda")
    d_corr = torch.empty(nf, dtype=torch.int32, device="cuda")
    chain.work_device(d_syms.data_ptr(), nf, d_n0.data_ptr(), 1, d_msg.data_ptr(), d_ret.data_ptr(), d_corr.data_ptr(),
                      torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    # oracle chain on the same inputs
    llr = T.oracle_demap(rx, n0, 8, 0)
    if T.ref_ldpc() is not None:
        dec_llr, wret = T.ref_ldpc_decode(fi["table"], llr, 0, 30)
    else:
        dec_llr, wret = T.oracle_ldpc_decode(fi["table"], llr, G, 30)
    want_msg, want_corr = ob.decode_bytes(T.pack_bits(dec_llr, fi["bch_n"]))
    assert d_ret.cpu().tolist() == wret
    assert d_corr.cpu().numpy().tolist() == want_corr.tolist()
    assert np.array_equal(d_msg.cpu().numpy(), want_msg)
    assert np.array_equal(want_msg[:nf - 1], msg[:nf - 1])  # the good frames decode to what was sent
    # same batch with bbdescrambler_bb fused into the BCH output stage
    chain.set_descramble(True)
    chain.work_device(d_syms.data_ptr(), nf, d_n0.data_ptr(), 1, d_msg.data_ptr(), d_ret.data_ptr(), d_corr.data_ptr(),
                      torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert d_corr.cpu().numpy().tolist() == want_corr.tolist()
    assert np.array_equal(d_msg.cpu().numpy(), T.oracle_bb_descramble(want_msg))
    chain.close()


# ------------------------------------------------------------------ LLR-domain chain (BASELINE config 5 shape, small batch)
@pytest.mark.parametrize("rate", ["C9_10", "C154_180"])
def test_chain_from_llrs(rate):
    """ldpc_decoder_bb -> bch_decoder_bb from int8 LLRs: 9/10 normal = DVB_S2_TABLE_B11 + BCH(58320, 58192, t = 8), and the
    genuine S2X table 154/180 = DVB_S2X_TABLE_B21 + BCH(55440, 55248, t = 12). Frames: decodable, one with a few bit
    errors left for the BCH code (LLR signs flipped hard), one hopeless (noise): vs genuine LDPC reference + BCH oracle."""
    import torch
    fi = get_fec_info(capi.STANDARD_DVBS2, capi.FECFRAME_NORMAL, rate)
    ob, _ = bch_pair(capi.FECFRAME_NORMAL, rate)
    nf, G, cap = 32, 32, 20
    rng = np.random.default_rng(91)
    msg = rng.integers(0, 256, (nf, fi["bch_k"] // 8), dtype=np.uint8)
    cw = T.ldpc_encode(fi["table"], np.unpackbits(ob.encode_bytes(msg), axis=1))
    N = cw.shape[1]
    llr = np.clip(np.rint((1.0 - 2.0 * cw) * 9.0 + rng.normal(0, 3.3, cw.shape)), -128, 127).astype(np.int8)
    llr[nf - 1] = T.llr_noise(1, N, 5)[0]
    chain = FecChain(rate=rate, group_size=G, max_frames=nf, max_trials=cap, from_llr=True)
    assert (chain.n_llr, chain.msg_bytes, chain.n_syms) == (N, fi["bch_k"] // 8, 0)
    d_llr = torch.from_numpy(llr).cuda()
    d_msg = torch.empty((nf, chain.msg_bytes), dtype=torch.uint8, device="cuda")
    d_ret = torch.empty(1, dtype=torch.int32, device="cuda")
    d_corr = torch.empty(nf, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    chain.work_llr_device(d_llr.data_ptr(), nf, d_msg.data_ptr(), d_ret.data_ptr(), d_corr.data_ptr(), st)
    if T.ref_ldpc() is not None:
        dec_llr, wret = T.ref_ldpc_decode(fi["table"], llr, 0, cap)
    else:
        dec_llr, wret = T.oracle_ldpc_decode(fi["table"], llr, G, cap)
    want_msg, want_corr = ob.decode_bytes(T.pack_bits(dec_llr, fi["bch_n"]))
    assert d_ret.cpu().tolist() == wret
    assert d_corr.cpu().numpy().tolist() == want_corr.tolist()
    assert np.array_equal(d_msg.cpu().numpy(), want_msg)
    assert np.array_equal(want_msg[:nf - 1], msg[:nf - 1]) and want_corr[nf - 1] < 0
    # enqueue / finish give the same bytes
    d_msg2 = torch.zeros_like(d_msg)
    chain.enqueue_llr_device(d_llr.data_ptr(), nf, d_msg2.data_ptr(), d_ret.data_ptr(), d_corr.data_ptr(), st)
    chain.finish()
    assert torch.equal(d_msg, d_msg2)
    chain.close()


# ------------------------------------------------------------------ host-pointer form of the fused chain (dvbs2_chain_decode)
def _chain_symbols(framesize, rate, constellation, nf, seed, es_n0_db, order=0):
    """Mapped BCH o LDPC codewords + AWGN (a few frames are pure noise), N0, and what was sent."""
    fi = get_fec_info(capi.STANDARD_DVBS2, framesize, rate)
    ob, _ = bch_pair(framesize, rate)
    rng = np.random.default_rng(seed)
    msg = rng.integers(0, 256, (nf, fi["bch_k"] // 8), dtype=np.uint8)
    cw = T.ldpc_encode(fi["table"], np.unpackbits(ob.encode_bytes(msg), axis=1))
    if constellation == capi.MOD_QPSK:
        syms = ((1 - 2.0 * cw[:, 0::2]) + 1j * (1 - 2.0 * cw[:, 1::2])) * np.sqrt(0.5)
    else:
        rows = cw.shape[1] // 3  # bit k of symbol s is codeword bit ra_k + s (the block's column de-interleaver, lib/xfecframe_demapper_cb_impl.cc:50-69,162-176)
        ra = {0: (0, rows, 2 * rows), 1: (2 * rows, rows, 0), 2: (rows, 0, 2 * rows)}[order]
        syms = T.map_8psk(np.stack([cw[:, a:a + rows] for a in ra], axis=-1))
    n0 = np.float32(10 ** (-es_n0_db / 10))
    noise = np.sqrt(n0 / 2) * (rng.normal(size=syms.shape) + 1j * rng.normal(size=syms.shape))
    rx = (syms + noise).astype(np.complex64)
    for f in range(3, nf, 17):
        rx[f] = (noise[f] * 3).astype(np.complex64)  # hopeless frames: their groups run to the cap, the BCH sees garbage
    return fi, ob, rx, n0, msg


@pytest.mark.parametrize("framesize,rate,constellation,nf,chunk,es_n0_db,order", [
    (capi.FECFRAME_NORMAL, "C3_4", capi.MOD_8PSK, 70, 32, 8.5, 0),    # demapper fused into the sweep load; chunks 32 + 32 + 6
    (capi.FECFRAME_SHORT, "C1_4", capi.MOD_QPSK, 200, 32, 0.5, 0),    # parity-in-records sweep kernel: demapper launch per chunk; 7 chunks > 4 slots
    (capi.FECFRAME_SHORT, "C3_5", capi.MOD_8PSK, 97, 0, 7.0, 1),      # the measured plan (one chunk here), column order 210
])
def test_chain_host_entry(framesize, rate, constellation, nf, chunk, es_n0_db, order, monkeypatch):
    """dvbs2_chain_decode (HOST symbols -> HOST message bytes, chunked over four streams) against the CPU chain -- demapper
    restatement -> genuine LDPC reference -> BCH codec --, with pageable and with page-locked caller buffers, one N0 and one per frame;
    and against the device-pointer entry."""
    import torch
    if chunk:
        monkeypatch.setenv("DVBS2_HOST_CHUNK", str(chunk))
    cap, G = 25, 32
    fi, ob, rx, n0, sent = _chain_symbols(framesize, rate, constellation, nf, 1234 + nf, es_n0_db, order)
    llr = T.oracle_demap(rx, n0, 4 if constellation == capi.MOD_QPSK else 8, order)  # (the checker takes the constellation SIZE)
    dec_llr, wret = T.cpu_ldpc_decode_ragged(fi["table"], llr, G, cap)
    want_msg, want_corr = ob.decode_bytes(T.pack_bits(dec_llr, fi["bch_n"]))
    assert (want_corr >= 0).sum() > nf // 2 and (want_corr < 0).any()
    good = want_corr >= 0
    assert np.array_equal(want_msg[good], sent[good])
    chain = FecChain(framesize=framesize, rate=rate, constellation=constellation, group_size=G, max_frames=nf + 7, max_trials=cap)
    # pageable caller buffers, one N0
    msg, ret, corr = chain.work(rx, n0)
    assert ret.tolist() == wret and corr.tolist() == want_corr.tolist() and np.array_equal(msg, want_msg)
    # one N0 per frame (the block before the first llr_pdu, lib/xfecframe_demapper_cb_impl.cc:128-149)
    msg2, ret2, corr2 = chain.work(rx, np.full(nf, n0, np.float32))
    assert ret2.tolist() == wret and corr2.tolist() == want_corr.tolist() and np.array_equal(msg2, want_msg)
    # page-locked caller buffers (copy engine addresses them directly), outputs optional
    hin = torch.from_numpy(rx.view(np.float32).reshape(nf, -1)).pin_memory()
    hn0 = torch.tensor([float(n0)], dtype=torch.float32).pin_memory()
    hmsg = torch.zeros((nf, chain.msg_bytes), dtype=torch.uint8).pin_memory()
    hcorr = torch.zeros(nf, dtype=torch.int32).pin_memory()
    assert capi.lib.dvbs2_host_is_page_locked(hin.data_ptr(), hin.numel() * 4) == 1
    chain.work_host_ptr(hin.data_ptr(), nf, hn0.data_ptr(), 1, hmsg.data_ptr(), 0, hcorr.data_ptr())
    assert np.array_equal(hmsg.numpy(), want_msg) and hcorr.numpy().tolist() == want_corr.tolist()
    # a shorter call on the same handle, then the device-pointer entry: same bytes
    dec33, wret33 = T.cpu_ldpc_decode_ragged(fi["table"], llr[:33], G, cap)  # (frame 32 is now a group of its own)
    want33, corr33 = ob.decode_bytes(T.pack_bits(dec33, fi["bch_n"]))
    msg3, ret3, corr3 = chain.work(rx[:33], n0)
    assert np.array_equal(msg3, want33) and ret3.tolist() == wret33 and corr3.tolist() == corr33.tolist()
    d_syms = hin.cuda()
    d_n0 = hn0.cuda()
    d_msg = torch.empty((nf, chain.msg_bytes), dtype=torch.uint8, device="cuda")
    chain.work_device(d_syms.data_ptr(), nf, d_n0.data_ptr(), 1, d_msg.data_ptr(), 0, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(d_msg.cpu().numpy(), want_msg)
    # bad arguments leave the handle usable
    assert capi.lib.dvbs2_chain_decode(chain._h, None, 4, hn0.data_ptr(), 1, cap, hmsg.data_ptr(), None, None) == capi.EINVAL
    assert capi.lib.dvbs2_chain_decode(chain._h, hin.data_ptr(), nf + 8, hn0.data_ptr(), 1, cap, hmsg.data_ptr(), None, None) == capi.ESIZE
    assert capi.lib.dvbs2_chain_decode(chain._h, hin.data_ptr(), nf, hn0.data_ptr(), 2, cap, hmsg.data_ptr(), None, None) == capi.EINVAL
    msg4, _, _ = chain.work(rx, n0, want_ret=False)
    assert np.array_equal(msg4, want_msg)
    chain.close()


def test_chain_host_entry_from_llrs(monkeypatch):
    """dvbs2_chain_decode_llr (HOST LLRs -> HOST message bytes) on 9/10 normal (BASELINE config 5's code), three chunks."""
    monkeypatch.setenv("DVBS2_HOST_CHUNK", "32")
    rate, nf, G, cap = "C9_10", 96, 32, 20  # (whole reference batches: the CPU chain runs the genuine AVX2 decoder)
    fi = get_fec_info(capi.STANDARD_DVBS2, capi.FECFRAME_NORMAL, rate)
    ob, _ = bch_pair(capi.FECFRAME_NORMAL, rate)
    rng = np.random.default_rng(92)
    sent = rng.integers(0, 256, (nf, fi["bch_k"] // 8), dtype=np.uint8)
    cw = T.ldpc_encode(fi["table"], np.unpackbits(ob.encode_bytes(sent), axis=1))
    llr = np.clip(np.rint((1.0 - 2.0 * cw) * 9.0 + rng.normal(0, 3.3, cw.shape)), -128, 127).astype(np.int8)
    llr[40] = T.llr_noise(1, cw.shape[1], 5)[0]
    want_msg, want_corr, wret, _ = T.chain_expect(fi["table"], fi["bch_n"], fi["bch_t"], capi.FECFRAME_NORMAL, llr, cap)
    chain = FecChain(rate=rate, group_size=G, max_frames=nf, max_trials=cap, from_llr=True)
    msg, ret, corr = chain.work_llr(llr)
    assert ret.tolist() == list(wret) and corr.tolist() == list(want_corr) and np.array_equal(msg, want_msg)
    assert capi.lib.dvbs2_chain_decode(chain._h, llr.ctypes.data, nf, None, 0, cap, msg.ctypes.data, None, None) == capi.EINVAL  # no demapper in this chain
    chain.close()


# ------------------------------------------------------------------ BASELINE configs 3 and 5 at full size, the benchmark's own input
@pytest.mark.parametrize("config", ["config3", "config5", "config5_s2x"])
def test_full_batch_chains_vs_reference(config):
    """4096 frames of bench.py's never-converging input (same generators and seeds) through the chains, EVERY frame against the
    CPU chain: (demapper restatement ->) genuine AVX2 LDPC on all host cores -> BCH codec; messages, BCH results per frame and LDPC
    return values per group. config3: 8PSK 3/4 normal from noise symbols (demapper fused into the sweep kernel's load, B7,
    BCH(48600,48408,12)); config5: 9/10 normal from LLRs (B11 + BCH(58320,58192,8)); config5_s2x: 154/180 (S2X B21 + BCH t = 12)."""
    import torch
    nf, G, cap = 4096, 32, 50
    rate = {"config3": "C3_4", "config5": "C9_10", "config5_s2x": "C154_180"}[config]
    fi = get_fec_info(capi.STANDARD_DVBS2, capi.FECFRAME_NORMAL, rate)
    st = torch.cuda.current_stream().cuda_stream
    d_ret = torch.empty(nf // G, dtype=torch.int32, device="cuda")
    d_corr = torch.empty(nf, dtype=torch.int32, device="cuda")
    if config == "config3":
        chain = FecChain(rate=rate, constellation=capi.MOD_8PSK, group_size=G, max_frames=nf, max_trials=cap)
        g = torch.Generator(device="cuda"); g.manual_seed(777)
        syms = torch.randn((nf, chain.n_syms * 2), generator=g, device="cuda") * 0.7071
        n0 = torch.tensor([1.0], dtype=torch.float32, device="cuda")
        d_msg = torch.empty((nf, chain.msg_bytes), dtype=torch.uint8, device="cuda")
        chain.work_device(syms.data_ptr(), nf, n0.data_ptr(), 1, d_msg.data_ptr(), d_ret.data_ptr(), d_corr.data_ptr(), st)
        llr = T.oracle_demap(syms.cpu().numpy().view(np.complex64), np.float32(1.0), 8, 0)
    else:
        chain = FecChain(rate=rate, group_size=G, max_frames=nf, max_trials=cap, from_llr=True)
        g = torch.Generator(device="cuda"); g.manual_seed(888)
        d_llr = torch.clamp(torch.round(torch.randn((nf, chain.n_llr), generator=g, device="cuda") * 8.0), -128, 127).to(torch.int8)
        d_msg = torch.empty((nf, chain.msg_bytes), dtype=torch.uint8, device="cuda")
        chain.work_llr_device(d_llr.data_ptr(), nf, d_msg.data_ptr(), d_ret.data_ptr(), d_corr.data_ptr(), st)
        llr = d_llr.cpu().numpy()
    if T.ref_ldpc() is None:   # no prebuilt reference here: the restatement on the last two groups
        sl = slice(nf - 64, nf)
        want_msg, want_corr, wret, _ = T.chain_expect(fi["table"], fi["bch_n"], fi["bch_t"], capi.FECFRAME_NORMAL, llr[sl], cap)
        assert d_ret[-2:].cpu().tolist() == wret and d_corr[sl].cpu().numpy().tolist() == want_corr.tolist()
        assert np.array_equal(d_msg[sl].cpu().numpy(), want_msg)
    else:
        want_msg, want_corr, wret, _ = T.chain_expect(fi["table"], fi["bch_n"], fi["bch_t"], capi.FECFRAME_NORMAL, llr, cap)
        assert d_ret.cpu().tolist() == wret
        assert d_corr.cpu().numpy().tolist() == want_corr.tolist()
        got = d_msg.cpu().numpy()
        bad = np.nonzero((got != want_msg).any(axis=1))[0]
        assert bad.size == 0, f"{config}: {bad.size} frames differ, first {bad[:8]}"
    chain.close()
