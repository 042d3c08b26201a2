"""GPU parity: libdvbs2_fec_hip LDPC (through the C ABI) vs the CPU checkers, bit-exact on decoded LLRs,
packed bits and per-group return values. Checkers: oracle/liboracle.so (plain-C restatement) and, when the
prebuilt oracle/_ref travelled with the repo, the genuine reference decoders (AVX2 batch = 32 frames,
generic batch = 16 frames)."""
import ctypes as C
import os

import numpy as np
import pytest

import fec_testlib as T
from dvbs2rx_amd import ldpc_table_names, LdpcDecoder, capi, get_fec_info

pytestmark = pytest.mark.gpu


def checker(table, llr, G, trials):
    """Genuine reference when it exists for this G, else the restatement."""
    r = T.ref_ldpc()
    if r is not None and G in (16, 32):
        return T.ref_ldpc_decode(table, llr, 0 if G == 32 else 2, trials)
    return T.oracle_ldpc_decode(table, llr, G, trials)


def run_gpu(table, llr, G, trials, outputmode=capi.OM_CODEWORD, message_bits=None):
    N, K, _, _ = T.ldpc_info(table)
    dec = LdpcDecoder(table=table, message_bits=message_bits or K, group_size=G, max_frames=llr.shape[0],
                      max_trials=trials, outputmode=outputmode)
    bits, out, ret = dec.work(llr, want_llr=True)
    dec.close()
    return bits, out, ret


def compare(table, llr, G, trials):
    N, K, _, _ = T.ldpc_info(table)
    bits, out, ret = run_gpu(table, llr, G, trials)
    want, wret = checker(table, llr, G, trials)
    assert ret.tolist() == wret, (table, G)
    bad = np.nonzero((out != want).any(axis=1))[0]
    assert bad.size == 0, f"{table} G={G}: LLR mismatch in frames {bad[:8]}"
    assert np.array_equal(bits, T.pack_bits(want, N))
    return ret


@pytest.mark.parametrize("table,trials", [("S2_TABLE_C1", 25), ("S2_TABLE_B4", 6), ("S2_TABLE_B7", 4),
                                          ("S2_TABLE_B11", 4), ("S2X_TABLE_B21", 4), ("S2X_TABLE_C8", 6),
                                          ("S2_TABLE_C10", 5), ("T2_TABLE_A3", 4)])
def test_never_converging(table, trials):
    N = T.ldpc_info(table)[0]
    ret = compare(table, T.llr_noise(32, N, 12345), 32, trials)
    assert ret.tolist() == [-1]


@pytest.mark.parametrize("table,amp,sigma", [("S2_TABLE_B4", 6, 5.2), ("S2_TABLE_B7", 8, 3.6),
                                             ("S2_TABLE_B11", 10, 2.75), ("S2_TABLE_C1", 5, 6.5),
                                             ("S2X_TABLE_B21", 10, 3.35)])
@pytest.mark.parametrize("G", [32, 16])
def test_near_threshold_groups(table, amp, sigma, G):
    """Frames converge after different numbers of updates: exercises the batch-coupled stopping rule."""
    llr, _ = T.llr_codeword_awgn(table, 64, 99, amp=amp, sigma=sigma)
    ret = compare(table, llr, G, 50)
    assert len(ret) == 64 // G


VARIANTS = {  # every build of the sweep kernel gives the same bits (environment overrides of the per-table policy)
    "policy": {},
    "pr-byte-records": {"DVBS2_PR_W1": "0"},                                                       # parity in records with two-dword records also for degree <= 4
    "pr-plain": {"DVBS2_PR_V2": "0"},                                                             # the two-dword-record kernel with its plain nodes everywhere
    "pr-packed": {"DVBS2_PR": "1", "DVBS2_PR_W1": "0", "DVBS2_PR_V2": "1"},                          # parity in records on every eligible table (normal frames too), packed nodes in the regular middle layers
    "classic": {"DVBS2_PR": "0", "DVBS2_DENSE": "0"},                                              # no parity-in-records / dense build
    "plain": {"DVBS2_PR": "0", "DVBS2_DENSE": "0", "DVBS2_HZ2": "0", "DVBS2_V2": "0", "DVBS2_SOLO": "0"},          # byte messages, scalar nodes, pair workgroups
    "packed-pair": {"DVBS2_PR": "0", "DVBS2_DENSE": "0", "DVBS2_HZ2": "0", "DVBS2_V2": "1", "DVBS2_SOLO": "0"},    # packed nodes, six-bit messages, pair workgroups
    "plain-solo": {"DVBS2_PR": "0", "DVBS2_DENSE": "0", "DVBS2_HZ2": "0", "DVBS2_V2": "0", "DVBS2_SOLO": "1"},     # scalar nodes, one frame per workgroup
    "packed-solo": {"DVBS2_PR": "0", "DVBS2_DENSE": "0", "DVBS2_HZ2": "0", "DVBS2_V2": "1", "DVBS2_SOLO": "1"},
    "packed-pair-plain-hazard": {"DVBS2_PR": "0", "DVBS2_DENSE": "0", "DVBS2_HZ2": "0", "DVBS2_V2": "1", "DVBS2_SOLO": "0", "DVBS2_V2P": "0"},  # hazard layers through the plain node (what a wave whose record does not fit the packed format runs)
    "heavy-hazard": {"DVBS2_PR": "0", "DVBS2_DENSE": "0", "DVBS2_HZ2": "1"},                       # twelve ordered entries, two-level walk (degree classes >= 12)
    "soft-barrier": {"DVBS2_PR": "0", "DVBS2_DENSE": "0", "DVBS2_HZ2": "0", "DVBS2_V2": "1", "DVBS2_SOLO": "0", "DVBS2_SOFT_BARRIER": "1"},  # per-frame software barriers
}


@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("table", ldpc_table_names())
def test_every_table_bit_exact(table, variant, monkeypatch):
    """All 57 DVB-S2 / S2X / T2 tables of the reference: never-converging input (fixed trip count)
    and noisy codewords (groups converge at different counts), bit-exact LLRs, bits and return values -- with the kernel
    build the library picks for the table and with every other build forced."""
    for k, v in VARIANTS[variant].items():
        monkeypatch.setenv(k, v)
    N = T.ldpc_info(table)[0]
    assert compare(table, T.llr_noise(32, N, 777), 32, 3).tolist() == [-1]
    llr, _ = T.llr_codeword_awgn(table, 32, 4242, amp=12, sigma=3.0)
    llr[5] = T.llr_noise(1, N, 778)[0]  # one hopeless frame keeps its group of 16 running to the cap
    ret = compare(table, llr, 16, 12)
    assert ret[0] == -1


def test_group_of_one_and_tail():
    table = "S2_TABLE_C1"
    llr, _ = T.llr_codeword_awgn(table, 40, 5, amp=5, sigma=6.5)
    compare(table, llr, 1, 30)
    # 40 frames with G = 32: trailing group of 8 frames is a group of its own
    bits, out, ret = run_gpu(table, llr, 32, 30)
    a, ra = T.oracle_ldpc_decode(table, llr[:32], 32, 30)
    b, rb = T.oracle_ldpc_decode(table, llr[32:], 8, 30)
    assert ret.tolist() == ra + rb
    assert np.array_equal(out, np.concatenate([a, b]))


def test_saturation_and_zero_llrs():
    table = "S2_TABLE_C4"
    N, K, _, _ = T.ldpc_info(table)
    rng = np.random.default_rng(3)
    sat = rng.choice(np.array([-128, -127, 127, 126, 0], np.int8), (32, N))
    compare(table, sat, 32, 8)
    compare(table, np.zeros((32, N), np.int8), 32, 3)
    cw_llr, _ = T.llr_codeword_awgn(table, 32, 4, amp=127, sigma=0.0)
    ret = compare(table, cw_llr, 32, 10)
    assert ret.tolist() == [10]  # clean codewords: zero updates


def test_output_modes_and_counters():
    table = "S2_TABLE_B4"
    N, K, _, _ = T.ldpc_info(table)
    llr, cw = T.llr_codeword_awgn(table, 32, 8, amp=8, sigma=3.0)
    dec = LdpcDecoder(rate="C1_2", framesize=capi.FECFRAME_NORMAL, group_size=32, max_frames=32,
                      max_trials=0, outputmode=capi.OM_MESSAGE)
    assert (dec.N, dec.K, dec.message_bits, dec.max_trials) == (64800, 32400, 32400, 25)
    bits, _, ret = dec.work(llr)
    assert bits.shape == (32, 4050)
    assert np.array_equal(np.unpackbits(bits, axis=1), cw[:, :K])
    assert dec.get_average_trials() == 25 - ret[0]
    dec.close()


def test_device_pointer_entry():
    import torch
    table = "S2_TABLE_C1"
    N, K, _, _ = T.ldpc_info(table)
    llr = T.llr_noise(64, N, 77)
    dec = LdpcDecoder(table=table, message_bits=K, group_size=32, max_frames=64, max_trials=10,
                      outputmode=capi.OM_MESSAGE)
    d_in = torch.from_numpy(llr).cuda()
    d_bits = torch.empty((64, K // 8), dtype=torch.uint8, device="cuda")
    d_llr = torch.empty((64, N), dtype=torch.int8, device="cuda")
    d_ret = torch.empty(2, dtype=torch.int32, device="cuda")
    dec.work_device(d_in.data_ptr(), 64, d_bits.data_ptr(), d_llr.data_ptr(), d_ret.data_ptr(),
                    torch.cuda.current_stream().cuda_stream)
    want, wret = T.oracle_ldpc_decode(table, llr, 32, 10)
    assert d_ret.cpu().tolist() == wret
    assert np.array_equal(d_llr.cpu().numpy(), want)
    assert np.array_equal(d_bits.cpu().numpy(), T.pack_bits(want, K))
    dec.close()


def test_every_table_of_the_reference():
    """All 57 parity tables (every kernel variant / degree case): 3 updates on never-converging input plus a clean
    codeword, against the genuine reference (the restatement for G=32 when oracle/_ref is absent: slower, identical). No table is skipped."""
    import json, os
    rows = json.load(open(os.path.join(T.ROOT, "tests", "golden", "fec_params.json")))["rows"]
    tables = sorted({r["table"] for r in rows})
    assert len(tables) == 57
    # without oracle/_ref the restatement stands in for every table (slower, same answers: it is pinned by the digests)
    for table in tables:
        N, K, _, _ = T.ldpc_info(table)
        x = T.llr_noise(32, N, 4242)
        clean, _ = T.llr_codeword_awgn(table, 1, 7, amp=20, sigma=0.0)
        x[31] = clean[0]
        bits, out, ret = run_gpu(table, x, 32, 3)
        want, wret = checker(table, x, 32, 3)
        assert ret.tolist() == wret, table
        assert np.array_equal(out, want), table


def test_reference_digests_on_gpu():
    """tests/golden/ldpc_golden.json (digests of the GENUINE reference, tools/gen_ldpc_golden.py): the SURVEY 8(c) grid -- five
    BASELINE tables x {clean codewords (no update), near threshold, never converging, saturating, all zero} x batch 32 (AVX2)
    and 16 (generic) -- plus the short / medium / T2 cases, compared by SHA-256 of decoded LLRs, packed bits and return codes.
    Needs nothing but the committed fixture."""
    import json, os
    cases = json.load(open(os.path.join(T.ROOT, "tests", "golden", "ldpc_golden.json")))
    assert len(cases) >= 38
    for case in cases:
        x = T.make_input(case["table"], case["kind"], case["n_frames"], **case["params"])
        assert T.sha(x) == case["input_sha256"], "input generator drifted"
        N = x.shape[1]
        for G in (32, 16):
            bits, out, ret = run_gpu(case["table"], x, G, case["trials"])
            want = case["results"][str(G)]
            assert ret.tolist() == want["ret"], (case["table"], case["kind"], G)
            assert T.sha(out) == want["llr_sha256"], (case["table"], case["kind"], G)
            assert T.sha(bits) == want["bits_sha256"], (case["table"], case["kind"], G)


def test_ragged_and_empty_batches():
    """1 frame, odd counts (the second half of the last pair workgroup is empty), zero frames, too many frames."""
    import ctypes as C
    table = "S2_TABLE_C1"
    N, K, _, _ = T.ldpc_info(table)
    llr, _ = T.llr_codeword_awgn(table, 7, 123, amp=5, sigma=6.4)
    dec = LdpcDecoder(table=table, message_bits=K, group_size=1, max_frames=7, max_trials=20, outputmode=capi.OM_CODEWORD)
    for nf in (1, 3, 7):
        bits, out, ret = dec.work(llr[:nf], want_llr=True)
        want, wret = T.oracle_ldpc_decode(table, llr[:nf], 1, 20)
        assert ret.tolist() == wret and np.array_equal(out, want) and np.array_equal(bits, T.pack_bits(want, N))
    assert capi.lib.dvbs2_ldpc_decode(dec._h, None, 0, 20, 1, None, None, None) == capi.OK
    buf = np.zeros((8, N), np.int8); ob = np.zeros((8, N // 8), np.uint8)
    assert capi.lib.dvbs2_ldpc_decode(dec._h, buf.ctypes.data, 8, 20, 0, ob.ctypes.data, None, None) == capi.ESIZE
    assert capi.lib.dvbs2_ldpc_decode(dec._h, buf.ctypes.data, 1, 0, 0, ob.ctypes.data, None, None) == capi.EINVAL
    dec.close()
    # one update only, and a cap the decoder never reaches
    for trials in (1, 200):
        bits, out, ret = run_gpu(table, llr[:4], 4, trials)
        want, wret = T.oracle_ldpc_decode(table, llr[:4], 4, trials)
        assert ret.tolist() == wret and np.array_equal(out, want)


@pytest.mark.parametrize("force", ["1", "0"])
def test_parity_in_records_variant(monkeypatch, force):
    """Both kernel variants on the tables where the library would pick either one (DVBS2_PR overrides the policy):
    "parity in records" (ldpc_kernel_pr.hpp, parity LLRs in registers / message records) and the classic kernel must
    give the same bits: near-threshold groups (resume passes), never-converging input, hazard layers, first / last layer."""
    monkeypatch.setenv("DVBS2_PR", force)
    for table, amp, sigma in (("S2_TABLE_B4", 6, 5.2), ("S2_TABLE_B1", 4, 6.6), ("S2_TABLE_B3", 5, 5.6), ("S2_TABLE_C1", 5, 6.5)):
        llr, _ = T.llr_codeword_awgn(table, 64, 99, amp=amp, sigma=sigma)
        compare(table, llr, 32, 50)
        compare(table, llr[:48], 16, 30)
        # 33 frames: the last pair workgroup holds one frame and an idle half; its group is a group of one
        bits, out, ret = run_gpu(table, llr[:33], 16, 30)
        a, ra = T.oracle_ldpc_decode(table, llr[:32], 16, 30)
        b, rb = T.oracle_ldpc_decode(table, llr[32:33], 1, 30)
        assert ret.tolist() == ra + rb and np.array_equal(out, np.concatenate([a, b]))
        compare(table, T.llr_noise(32, T.ldpc_info(table)[0], 5), 32, 4)


def test_kernel_variant_policy(monkeypatch):
    """Which sweep kernel a handle launches (dvbs2_ldpc_kernel_name): parity-in-records for short/medium frames with check degree
    <= 7, the 80-VGPR build for hazard-dominated short tables, otherwise the build of the table's degree class that measured
    fastest (packed nodes and / or one frame per workgroup)."""
    def name(table, **env):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        K = T.ldpc_info(table)[1]
        dec = LdpcDecoder(table=table, message_bits=K, group_size=32, max_frames=32, max_trials=5)
        n = dec.kernel_name
        dec.close()
        for k in env:
            monkeypatch.delenv(k)
        return n
    # the build per table is data: csrc/ldpc_policy.inc (generated from an MI355X sweep of the four builds, tools/gen_policy.py)
    import re
    pol = {m.group(1): (int(m.group(2)), int(m.group(3))) for m in
           re.finditer(r'\{ "(\w+)", (\d), (\d) \}', open(os.path.join(T.ROOT, "gr-dvbs2rx_amd", "csrc", "ldpc_policy.inc")).read())}
    assert len(pol) == 57
    def expect(table, dmax):
        packed, solo = pol[table]
        solo = solo and dmax <= 16
        return f"ldpc_layered_kernel<{dmax}" + ((", packed, solo>" if packed else ", solo>") if solo else (", packed>" if packed else ">"))
    assert name("S2_TABLE_B4") == expect("S2_TABLE_B4", 8)
    assert name("S2_TABLE_B7") == expect("S2_TABLE_B7", 16)
    assert name("S2_TABLE_B11") == expect("S2_TABLE_B11", 32)
    assert name("S2X_TABLE_B9") == expect("S2X_TABLE_B9", 16)
    assert name("S2_TABLE_C2") == "ldpc_layered_pr_kernel<packed>"  # short / medium frames of degree <= 7: parity in records, packed nodes in the regular middle layers (round 6)
    assert name("S2_TABLE_C2", DVBS2_PR_V2="0") == "ldpc_layered_pr_kernel"
    assert name("S2_TABLE_C1") == "ldpc_layered_pr_kernel<w1>"    # ... of degree <= 4: one-dword records (6-bit messages + parity byte)
    assert name("S2_TABLE_C1", DVBS2_PR_W1="0") == "ldpc_layered_pr_kernel<packed>"  # (two-dword records: the packed build, whose packed nodes start at degree 5 -- none of this table's layers)
    assert name("S2_TABLE_C1", DVBS2_PR_W1="0", DVBS2_PR_V2="0") == "ldpc_layered_pr_kernel"
    assert name("S2X_TABLE_C9") == "ldpc_layered_pr_kernel<w1>"   # medium frame
    assert name("S2_TABLE_C5") == "ldpc_layered_kernel<12, dense>"  # short 3/5: 16 of 18 layers are hazard layers
    assert name("S2_TABLE_C5", DVBS2_DENSE="0", DVBS2_V2="0", DVBS2_SOLO="0") == "ldpc_layered_kernel<12>"
    assert name("S2_TABLE_C1", DVBS2_PR="0", DVBS2_V2="0", DVBS2_SOLO="0") == "ldpc_layered_kernel<4>"  # degree <= 4: one message dword per check
    assert name("S2_TABLE_B1") == "ldpc_layered_pr_kernel<w1>"       # 1/4 normal: the NORMAL tables of degree <= 4 take the one-dword-record kernel too (round 6, by rule)
    assert name("S2X_TABLE_B1") == "ldpc_layered_pr_kernel<w1>"      # S2X 2/9 normal likewise
    assert name("S2_TABLE_B1", DVBS2_PR="0") == expect("S2_TABLE_B1", 4)
    assert name("S2_TABLE_B2") == expect("S2_TABLE_B2", 8)           # 1/3 normal (degree 5): classic kernel
    assert name("S2_TABLE_B4", DVBS2_V2="1", DVBS2_SOLO="1") == "ldpc_layered_kernel<8, packed, solo>"
    assert name("S2_TABLE_B4", DVBS2_V2="0", DVBS2_SOLO="0") == "ldpc_layered_kernel<8>"
    assert name("S2_TABLE_B11", DVBS2_V2="1", DVBS2_SOLO="1") == "ldpc_layered_kernel<32, packed>"  # no one-frame build above 128 VGPRs
    assert name("S2X_TABLE_B21") == ("ldpc_layered_kernel<32, packed, soft>" if pol["S2X_TABLE_B21"][0] else "ldpc_layered_kernel<32, soft>")  # long layers, no hazard layer: per-frame software barriers
    assert name("S2_TABLE_C10") == "ldpc_layered_kernel<28, hz2>"     # short 9/10: ten and twelve ordered entries per check
    assert name("S2_TABLE_B9") == expect("S2_TABLE_B9", 24)           # 5/6 normal: hazard layers -> hardware barriers (ldpc_policy_soft.inc lists the exceptions: none now)
    assert name("S2_TABLE_B9", DVBS2_SOFT_BARRIER="1") == expect("S2_TABLE_B9", 24)[:-1] + ", soft>"
    assert name("S2_TABLE_B4", DVBS2_PR="1") == "ldpc_layered_pr_kernel"
    assert name("S2_TABLE_B1", DVBS2_PR="1") == "ldpc_layered_pr_kernel<w1>"


@pytest.mark.parametrize("table,nf,trials,amp,sigma", [("S2_TABLE_B4", 4096, 50, 6, 4.6), ("S2_TABLE_C1", 16384, 25, 5, 4.0)])
def test_full_batch_properties(table, nf, trials, amp, sigma):
    """BASELINE.json batch sizes (4096 normal / 16384 short frames per GPU), checked through size-independent properties:
    encode -> noise -> decode round trip for every frame, idempotence (decoding the decoded LLRs needs 0 updates and
    changes nothing), independence of the batch size (a 64-frame slice decoded alone gives the same bytes and group
    results), and that slice against the CPU reference."""
    N, K, _, _ = T.ldpc_info(table)
    rng = np.random.default_rng(2024)
    base = 64
    info = rng.integers(0, 2, (base, K), dtype=np.uint8)
    cw = T.ldpc_encode(table, info)
    reps = nf // base
    y = amp * (1.0 - 2.0 * cw.astype(np.float32))
    llr = np.empty((nf, N), np.int8)
    for r in range(reps):  # same codewords, fresh noise per repetition
        llr[r * base:(r + 1) * base] = np.clip(np.rint(y + sigma * rng.standard_normal((base, N), dtype=np.float32)), -128, 127)
    dec = LdpcDecoder(table=table, message_bits=K, group_size=32, max_frames=nf, max_trials=trials, outputmode=capi.OM_CODEWORD)
    bits, out, ret = dec.work(llr, want_llr=True)
    assert (ret >= 0).all() and ret.max() < trials          # every group converged after at least one update
    want_bits = np.packbits(cw, axis=1)
    assert np.array_equal(bits.reshape(reps, base, -1), np.broadcast_to(want_bits, (reps, base, N // 8)))
    bits2, out2, ret2 = dec.work(out, want_llr=True)          # idempotence
    assert (ret2 == trials).all() and np.array_equal(bits2, bits) and np.array_equal(out2, out)
    lo = 5 * base
    b3, o3, r3 = dec.work(llr[lo:lo + base], want_llr=True)   # batch-size independence
    assert np.array_equal(b3, bits[lo:lo + base]) and np.array_equal(o3, out[lo:lo + base])
    assert r3.tolist() == ret[lo // 32:(lo + base) // 32].tolist()
    want, wret = checker(table, llr[lo:lo + base], 32, trials)
    assert np.array_equal(o3, want) and r3.tolist() == wret
    dec.close()


def test_async_entry_and_chunked_host_path(monkeypatch):
    """dvbs2_ldpc_enqueue_device + dvbs2_ldpc_finish == dvbs2_ldpc_decode_device; the host entry == the device entry (416 frames
    are ONE chunk of the default host pipeline: chunk = max(512, n_frames / 8); the multi-chunk pipeline is
    test_host_entry_multi_chunk); and with no device-side resolution rounds enqueued (DVBS2_RESOLVE_ROUNDS=0) the
    host-side leftover rounds of finish() give the same result (near-threshold input: groups need resume passes)."""
    import torch
    table = "S2_TABLE_C1"
    N, K, _, _ = T.ldpc_info(table)
    nf, G, cap = 416, 32, 25  # 13 groups
    llr, _ = T.llr_codeword_awgn(table, nf, 2025, amp=5, sigma=6.0)
    want, wret = checker(table, llr, G, cap)
    assert len(set(wret)) > 2  # groups stop at different counts: the stopping rule is exercised
    st = torch.cuda.current_stream().cuda_stream
    for rounds in (None, "0"):
        if rounds is not None:
            monkeypatch.setenv("DVBS2_RESOLVE_ROUNDS", rounds)
        dec = LdpcDecoder(table=table, message_bits=K, group_size=G, max_frames=nf, max_trials=cap, outputmode=capi.OM_MESSAGE)
        d_in = torch.from_numpy(llr).cuda()
        d_bits = torch.zeros((nf, K // 8), dtype=torch.uint8, device="cuda")
        d_out = torch.zeros((nf, N), dtype=torch.int8, device="cuda")
        d_ret = torch.zeros(nf // G, dtype=torch.int32, device="cuda")
        dec.enqueue_device(d_in.data_ptr(), nf, d_bits.data_ptr(), d_out.data_ptr(), d_ret.data_ptr(), st)
        dec.finish()
        assert d_ret.cpu().tolist() == wret
        assert np.array_equal(d_out.cpu().numpy(), want)
        assert np.array_equal(d_bits.cpu().numpy(), T.pack_bits(want, K))
        bits, out, ret = dec.work(llr, want_llr=True)  # host buffers, chunked
        assert ret.tolist() == wret and np.array_equal(out, want) and np.array_equal(bits, T.pack_bits(want, K))
        dec.close()


@pytest.mark.parametrize("framesize,rate", [(capi.FECFRAME_SHORT, "C1_5_VLSNR_SF2"), (capi.FECFRAME_MEDIUM, "C1_5_MEDIUM")])
def test_rows_whose_bch_length_is_not_the_table_k(framesize, rate):
    """VL-SNR / medium rows of get_fec_info() where bch.n != K of the parity table (lib/fec_params.cc:291-295, :323-327:
    2680 vs 3240 on DVB_S2_TABLE_C1, 5840 vs 6480 on DVB_S2X_TABLE_C8): created by (standard, framesize, rate) like the
    block does, OM_MESSAGE emits the first ldpc.k = bch.n bits of each decoded frame (lib/ldpc_decoder_bb_impl.cc:432-442),
    the decoder itself works on all N LLRs. Against the genuine reference decoder."""
    fi = get_fec_info(capi.STANDARD_DVBS2, framesize, rate)
    N, K, _, _ = T.ldpc_info(fi["table"])
    assert fi["ldpc_k"] != K and fi["ldpc_n"] == N and fi["ldpc_k"] == fi["bch_n"]
    dec = LdpcDecoder(standard=capi.STANDARD_DVBS2, framesize=framesize, rate=rate, outputmode=capi.OM_MESSAGE,
                      max_trials=20, group_size=32, max_frames=64)
    assert (dec.N, dec.K, dec.message_bits, dec.out_bytes) == (N, K, fi["ldpc_k"], fi["ldpc_k"] // 8)
    llr, _ = T.llr_codeword_awgn(fi["table"], 64, 311, amp=5, sigma=6.0)
    llr[40] = T.llr_noise(1, N, 9)[0]
    bits, out, ret = dec.work(llr, want_llr=True)
    want, wret = checker(fi["table"], llr, 32, 20)
    assert ret.tolist() == wret and np.array_equal(out, want)
    assert np.array_equal(bits, T.pack_bits(want, fi["ldpc_k"]))
    dec.close()


def test_enqueue_finish_contract():
    """One call may be outstanding per handle: a second enqueue before finish() is refused (and does not disturb the first), finish()
    without a call is a no-op, a failed call does not poison the handle, results of the first call are complete after its finish()."""
    import torch
    from dvbs2rx_amd.capi import Dvbs2Error
    table = "S2_TABLE_C1"
    N, K, _, _ = T.ldpc_info(table)
    nf, G, cap = 64, 32, 25
    llr, _ = T.llr_codeword_awgn(table, nf, 5, amp=5, sigma=6.0)
    want, wret = checker(table, llr, G, cap)
    st = torch.cuda.current_stream().cuda_stream
    dec = LdpcDecoder(table=table, message_bits=K, group_size=G, max_frames=nf, max_trials=cap, outputmode=capi.OM_MESSAGE)
    dec.finish()  # nothing outstanding
    d_in = torch.from_numpy(llr).cuda()
    d_bits = torch.zeros((nf, K // 8), dtype=torch.uint8, device="cuda")
    d_ret = torch.zeros(nf // G, dtype=torch.int32, device="cuda")
    dec.enqueue_device(d_in.data_ptr(), nf, d_bits.data_ptr(), 0, d_ret.data_ptr(), st)
    with pytest.raises(Dvbs2Error):
        dec.enqueue_device(d_in.data_ptr(), nf, d_bits.data_ptr(), 0, d_ret.data_ptr(), st)
    dec.finish()
    assert d_ret.cpu().tolist() == wret and np.array_equal(d_bits.cpu().numpy(), T.pack_bits(want, K))
    with pytest.raises(Dvbs2Error):  # more frames than the handle was made for
        dec.work_device(d_in.data_ptr(), nf + 32, d_bits.data_ptr(), 0, d_ret.data_ptr(), st)
    d_bits.zero_()
    dec.work_device(d_in.data_ptr(), nf, d_bits.data_ptr(), 0, d_ret.data_ptr(), st)  # the handle still works
    assert np.array_equal(d_bits.cpu().numpy(), T.pack_bits(want, K))
    dec.close()


@pytest.mark.parametrize("nf,chunk,rounds", [(416, "64", None), (416, "64", "0"), (1088, None, None), (1088, None, "0"), (200, "66", "0")])
def test_host_entry_multi_chunk(monkeypatch, nf, chunk, rounds):
    """The chunked pipeline of the host-buffer entry dvbs2_ldpc_decode (csrc/c_api.hip): frame_base > 0, all four slots and
    streams, slot re-use after finish(c - kSlots) (seven chunks of 64 frames), the pinned landing buffers, and -- with no
    device-side resolution rounds (DVBS2_RESOLVE_ROUNDS=0) on near-threshold input -- the re-fetch of a chunk's outputs when
    finish() had to run host-driven rounds. 1088 frames take the DEFAULT chunking (512 + 512 + 64); chunk 66 with G = 32 is
    rounded up to whole groups (96). Against the genuine reference on the whole batch."""
    if chunk is not None:
        monkeypatch.setenv("DVBS2_HOST_CHUNK", chunk)
    if rounds is not None:
        monkeypatch.setenv("DVBS2_RESOLVE_ROUNDS", rounds)
    table, G, cap = "S2_TABLE_C1", 32, 25
    N, K, _, _ = T.ldpc_info(table)
    base, _ = T.llr_codeword_awgn(table, 96, 2026, amp=5, sigma=6.0)
    llr = np.tile(base, ((nf + 95) // 96, 1))[:nf].copy()
    llr[nf // 2] = T.llr_noise(1, N, 4)[0]              # one hopeless frame: its group runs to the cap
    tail = nf % G
    want, wret = checker(table, llr[:nf - tail], G, cap)
    if tail:
        w2, r2 = T.oracle_ldpc_decode(table, llr[nf - tail:], tail, cap)
        want, wret = np.concatenate([want, w2]), wret + r2
    assert len(set(wret)) > 2
    dec = LdpcDecoder(table=table, message_bits=K, group_size=G, max_frames=nf, max_trials=cap, outputmode=capi.OM_MESSAGE)
    for _ in range(2):                                   # the second call re-uses every slot, stream and landing buffer
        bits, out, ret = dec.work(llr, want_llr=True)
        assert ret.tolist() == wret
        assert np.array_equal(out, want) and np.array_equal(bits, T.pack_bits(want, K))
    bits, _, ret = dec.work(llr)                         # without the soft output (no llr landing buffer in the copies)
    assert ret.tolist() == wret and np.array_equal(bits, T.pack_bits(want, K))
    dec.close()


def test_baseline_config1_one_frame_replicated():
    """BASELINE.json config 1 ("QPSK 1/2 normal FECFRAME, 50 iters, 1 frame ... plumbing"), SURVEY 8(d) row 1: ONE frame
    replicated over the 32 lanes of a reference batch -- all lanes identical, so group semantics = single-frame semantics --
    through the host-buffer entry dvbs2_ldpc_decode (what ldpc_decoder_bb's general_work would call,
    lib/ldpc_decoder_bb_impl.cc:406-449), created by (standard, framesize, rate) like the block, cap 50; decoded LLRs, packed
    message bits and the return value against the genuine AVX2 decoder (lib/ldpc_decoder/layered_decoder.hh:143-160).
    Three frames: one that converges after several updates, a clean codeword (no update), noise (runs the cap, ret -1)."""
    table = "S2_TABLE_B4"
    N, K, _, _ = T.ldpc_info(table)
    near, cw = T.llr_codeword_awgn(table, 1, 31, amp=6, sigma=5.0)
    clean, _ = T.llr_codeword_awgn(table, 1, 32, amp=20, sigma=0.0)
    dec = LdpcDecoder(standard=capi.STANDARD_DVBS2, framesize=capi.FECFRAME_NORMAL, rate="C1_2", outputmode=capi.OM_MESSAGE,
                      max_trials=50, group_size=32, max_frames=32)
    assert dec.kernel_name.startswith("ldpc_layered")
    seen = []
    for frame in (near[0], clean[0], T.llr_noise(1, N, 33)[0]):
        x = np.tile(frame, (32, 1))
        bits, out, ret = dec.work(x, want_llr=True)
        want, wret = checker(table, x, 32, 50)
        assert ret.tolist() == wret
        assert np.array_equal(out, want) and np.array_equal(bits, T.pack_bits(want, K))
        assert (out == out[0]).all()                     # identical lanes stay identical
        seen.append(wret[0])
    assert 0 < seen[0] < 50 and seen[1] == 50 and seen[2] == -1, seen
    assert np.array_equal(np.unpackbits(dec.work(np.tile(near[0], (32, 1)))[0][0]), cw[0, :K])
    dec.close()


def _torch_noise(nf, N, seed):
    import torch
    g = torch.Generator(device="cuda"); g.manual_seed(seed)
    return torch.clamp(torch.round(torch.randn((nf, N), generator=g, device="cuda") * 8.0), -128, 127).to(torch.int8)


@pytest.mark.parametrize("table,nf,trials,seed", [("S2_TABLE_B4", 4096, 50, 12345), ("S2_TABLE_C1", 16384, 25, 777)])
def test_full_batch_of_the_benchmark_input_vs_reference(table, nf, trials, seed):
    """BASELINE configs 2 and 4 at FULL size on the benchmark's own never-converging input (bench.py: same generator, same seed),
    EVERY frame against the genuine AVX2 reference run on all host cores (tests/fec_testlib.ref_ldpc_decode_parallel): decoded
    LLRs, packed bits, return values. Covers what the first-group gates cannot: high workgroup indices, the second half of every
    pair workgroup, the one-frame builds' CU counters, several workgroups per CU in sequence."""
    import torch
    N, K, _, _ = T.ldpc_info(table)
    d_in = _torch_noise(nf, N, seed)
    dec = LdpcDecoder(table=table, message_bits=K, group_size=32, max_frames=nf, max_trials=trials, outputmode=capi.OM_MESSAGE)
    d_bits = torch.empty((nf, K // 8), dtype=torch.uint8, device="cuda")
    d_llr = torch.empty((nf, N), dtype=torch.int8, device="cuda")
    d_ret = torch.empty(nf // 32, dtype=torch.int32, device="cuda")
    dec.work_device(d_in.data_ptr(), nf, d_bits.data_ptr(), d_llr.data_ptr(), d_ret.data_ptr(), torch.cuda.current_stream().cuda_stream)
    x = d_in.cpu().numpy()
    if T.ref_ldpc() is not None:
        want, wret = T.ref_ldpc_decode_parallel(table, x, 0, trials)
    else:                                                # no prebuilt reference here: the restatement on a slice of the batch
        sl = slice(nf - 64, nf)
        want, wret = T.oracle_ldpc_decode(table, x[sl], 32, trials)
        assert np.array_equal(d_llr[sl].cpu().numpy(), want) and d_ret[-2:].cpu().tolist() == wret
        dec.close()
        return
    assert d_ret.cpu().tolist() == wret == [-1] * (nf // 32)
    got = d_llr.cpu().numpy()
    bad = np.nonzero((got != want).any(axis=1))[0]
    assert bad.size == 0, f"{table}: {bad.size} frames differ, first {bad[:8]}"
    assert np.array_equal(d_bits.cpu().numpy(), T.pack_bits(want, K))
    dec.close()


def test_cu_slot_table_is_per_device():
    """The per-device table of the one-frame kernels' wave-pattern counters (csrc/ldpc_hip.hip, cu_slot_table), exercised with device KEYS
    a one-GPU box does not have: one array per key, the same array for the same key, zeroed, and the array of the real device is the one
    the handles use (its counters are back to zero once a decode has finished)."""
    import ctypes as C
    addr, nz = C.c_ulonglong(), C.c_int()
    seen = {}
    for key in (0, 1, 7, 1, 0, 7):
        capi.check(capi.lib.dvbs2_debug_cu_slot_table(0, key, C.byref(addr), C.byref(nz)))
        assert nz.value == 0 and addr.value != 0
        assert seen.setdefault(key, addr.value) == addr.value
    assert len(set(seen.values())) == 3
    table = "S2_TABLE_B4"  # one-frame packed build
    N, K, _, _ = T.ldpc_info(table)
    dec = LdpcDecoder(table=table, message_bits=K, group_size=32, max_frames=64, max_trials=5, outputmode=capi.OM_MESSAGE)
    assert "solo" in dec.kernel_name
    dec.work(T.llr_noise(64, N, 3))
    capi.check(capi.lib.dvbs2_debug_cu_slot_table(0, 0, C.byref(addr), C.byref(nz)))
    assert addr.value == seen[0] and nz.value == 0  # every workgroup gave its pattern slot back
    dec.close()
    assert capi.lib.dvbs2_debug_cu_slot_table(0, 0, None, None) == capi.EINVAL


def test_two_devices_from_one_process():
    """one host thread + stream per GPU. Two handles on devices 0 and 1 driven from ONE process (the caller's
    current device stays where it was, csrc/device_guard.h), each decoding its contiguous G-aligned share; together they equal
    the unsharded decode. Skips on a box with fewer than two GPUs."""
    import torch
    if capi.lib.dvbs2_device_count() < 2:
        pytest.skip("needs two HIP devices")
    table, G, cap = "S2_TABLE_C1", 32, 25
    N, K, _, _ = T.ldpc_info(table)
    llr, _ = T.llr_codeword_awgn(table, 128, 77, amp=5, sigma=6.0)
    want, wret = checker(table, llr, G, cap)
    torch.cuda.set_device(0)
    decs = [LdpcDecoder(table=table, message_bits=K, group_size=G, max_frames=64, max_trials=cap, outputmode=capi.OM_MESSAGE, device=d)
            for d in (0, 1)]
    assert torch.cuda.current_device() == 0              # create() on device 1 did not move the caller
    bufs = []
    for d, dec in enumerate(decs):
        dev = torch.device("cuda", d)
        x = torch.from_numpy(llr[64 * d:64 * d + 64]).to(dev)
        b = torch.zeros((64, K // 8), dtype=torch.uint8, device=dev)
        o = torch.zeros((64, N), dtype=torch.int8, device=dev)
        r = torch.zeros(2, dtype=torch.int32, device=dev)
        st = torch.cuda.Stream(device=dev)
        dec.enqueue_device(x.data_ptr(), 64, b.data_ptr(), o.data_ptr(), r.data_ptr(), st.cuda_stream)  # both GPUs busy before either is waited for
        bufs.append((x, b, o, r, st))
    assert torch.cuda.current_device() == 0
    for d, dec in enumerate(decs):
        dec.finish()
        x, b, o, r, st = bufs[d]
        assert r.cpu().tolist() == wret[2 * d:2 * d + 2]
        assert np.array_equal(o.cpu().numpy(), want[64 * d:64 * d + 64])
        assert np.array_equal(b.cpu().numpy(), T.pack_bits(want[64 * d:64 * d + 64], K))
        dec.close()


@pytest.mark.parametrize("table,amp,sigma", [("S2_TABLE_B4", 6, 5.2), ("S2_TABLE_C1", 5, 6.0)])
def test_group_stop_fallback_when_members_give_up(monkeypatch, table, amp, sigma):
    """The group-synchronous stop (csrc/ldpc_kernel.hpp, group_decide) lets a frame that passes its test wait for the other frames of
    its group; if they do not report in time it stops at its own good point and the host-side resolution (targets kernel + resume
    launches, finish()) completes the group. DVBS2_GROUP_SPIN_MAX=0 makes every waiting frame give up at once: near-threshold
    groups (frames converge at different counts) must still come out exactly like the reference's lockstep batch, for G = 32 and 16,
    with and without enqueued resolution rounds."""
    monkeypatch.setenv("DVBS2_GROUP_SPIN_MAX", "0")
    llr, _ = T.llr_codeword_awgn(table, 96, 99, amp=amp, sigma=sigma)
    llr[70] = T.llr_noise(1, llr.shape[1], 3)[0]
    for rounds in ("2", "0"):
        monkeypatch.setenv("DVBS2_RESOLVE_ROUNDS", rounds)
        r32 = compare(table, llr, 32, 50)
        compare(table, llr[:80], 16, 30)
        assert len(set(r32.tolist())) > 1


def test_group_stop_needs_no_host_round_at_the_default_threshold():
    """a member that gives up waiting for its group is repaired by host-driven rounds (finish()); at the default
    give-up threshold that must not happen -- the library counts those rounds (dvbs2_ldpc_fallback_rounds) and a converging batch of
    the headline table, decoded several times through the device and the chunked host entry, must leave the counter at zero."""
    table = "S2_TABLE_B4"
    N, K, _, _ = T.ldpc_info(table)
    llr, _ = T.llr_codeword_awgn(table, 1024, 77, amp=6, sigma=4.8)
    dec = LdpcDecoder(table=table, message_bits=K, group_size=32, max_frames=1024, max_trials=50, outputmode=capi.OM_MESSAGE)
    want = None
    for _ in range(4):
        bits, _, ret = dec.work(llr, want_llr=False)
        if want is None:
            want = (bits.copy(), ret.copy())
        assert np.array_equal(bits, want[0]) and np.array_equal(ret, want[1])
    assert len(set(ret.tolist())) > 1 and (ret >= 0).all()
    assert dec.fallback_rounds == 0
    dec.close()


def test_two_handles_pipelined_on_two_streams():
    """What a double-buffering block does (bench.py, config2_awgn.pipelined): two handles, enqueue / finish on one
    stream each, call i finished right before call i + 2 is enqueued, so two sweep kernels share the GPU most of the time. Converging
    groups (the group-synchronous stop waits for co-resident members: workgroups of two launches are dispatched interleaved) must come
    out like the reference's lockstep batches on every call, and no member may have given up waiting (fallback rounds stay zero)."""
    import torch
    table = "S2_TABLE_B4"
    N, K, _, _ = T.ldpc_info(table)
    nf = 2048
    llrs = [T.llr_codeword_awgn(table, nf, 500 + i, amp=6, sigma=4.8)[0] for i in range(2)]
    wants = [T.ref_ldpc_decode_parallel(table, x, 0, 50) if T.ref_ldpc() is not None else T.oracle_ldpc_decode(table, x[:64], 32, 50) for x in llrs]
    decs = [LdpcDecoder(table=table, message_bits=K, group_size=32, max_frames=nf, max_trials=50, outputmode=capi.OM_MESSAGE) for _ in range(2)]
    d_in = [torch.from_numpy(x).cuda() for x in llrs]
    d_bits = [torch.zeros((nf, K // 8), dtype=torch.uint8, device="cuda") for _ in range(2)]
    d_out = [torch.zeros((nf, N), dtype=torch.int8, device="cuda") for _ in range(2)]
    d_ret = [torch.zeros(nf // 32, dtype=torch.int32, device="cuda") for _ in range(2)]
    sts = [torch.cuda.Stream(), torch.cuda.Stream()]
    torch.cuda.synchronize()
    for i in range(12):
        h = i % 2
        if i >= 2:
            decs[h].finish()
            nchk = len(wants[h][1])
            assert d_ret[h].cpu().tolist()[:nchk] == wants[h][1], (i, h)
            assert np.array_equal(d_out[h][:32 * nchk].cpu().numpy(), wants[h][0][:32 * nchk]), (i, h)
            assert np.array_equal(d_bits[h][:32 * nchk].cpu().numpy(), T.pack_bits(wants[h][0][:32 * nchk], K)), (i, h)
            d_ret[h].zero_(); d_out[h].zero_(); torch.cuda.synchronize()
        decs[h].enqueue_device(d_in[h].data_ptr(), nf, d_bits[h].data_ptr(), d_out[h].data_ptr(), d_ret[h].data_ptr(), sts[h].cuda_stream)
    for h in range(2):
        decs[h].finish()
        assert d_ret[h].cpu().tolist()[:len(wants[h][1])] == wants[h][1]
    assert len(set(wants[0][1])) > 1  # groups stop at different counts
    assert decs[0].fallback_rounds == 0 and decs[1].fallback_rounds == 0
    for d in decs:
        d.close()


def test_host_link_measurement_entry():
    """dvbs2_measure_host_copy (diagnostics of bench.py's host_link): argument checks and a plausible rate for every kind of host memory."""
    import ctypes as C
    h2d, d2h = C.c_double(), C.c_double()
    assert capi.lib.dvbs2_measure_host_copy(0, 0, 1, 0, h2d, d2h) == capi.EINVAL
    assert capi.lib.dvbs2_measure_host_copy(0, 1 << 20, 0, 0, h2d, d2h) == capi.EINVAL
    assert capi.lib.dvbs2_measure_host_copy(0, 1 << 20, 1, 3, h2d, d2h) == capi.EINVAL
    for kind, streams in ((0, 1), (0, 4), (1, 1), (2, 1)):
        capi.check(capi.lib.dvbs2_measure_host_copy(0, 64 << 20, streams, kind, h2d, d2h))
        assert 0.2 < h2d.value < 200.0 and 0.2 < d2h.value < 200.0, (kind, streams, h2d.value, d2h.value)


def test_host_range_page_locked_predicate():
    """the host entry lets the copy engine address a caller's buffer directly only when the WHOLE range lies inside one
    page-locked allocation / registration as the runtime records it (dvbs2_host_is_page_locked = the predicate dvbs2_ldpc_decode
    applies). Pageable memory, a range that runs past its registration, and a range spanning TWO registrations with a pageable hole
    between them must all be refused (they go through the handle's pinned buffers instead)."""
    import ctypes as C
    import mmap
    import torch
    page = mmap.PAGESIZE
    buf = mmap.mmap(-1, 64 * page)
    base = C.addressof(C.c_char.from_buffer(buf))
    assert capi.lib.dvbs2_host_is_page_locked(base, 4 * page) == 0                      # pageable
    capi.check(capi.lib.dvbs2_host_register(base, 8 * page))                            # pages 0..7
    capi.check(capi.lib.dvbs2_host_register(base + 9 * page, 8 * page))                 # pages 9..16 (page 8 stays pageable)
    try:
        assert capi.lib.dvbs2_host_is_page_locked(base, 8 * page) == 1
        assert capi.lib.dvbs2_host_is_page_locked(base + page, 3 * page + 17) == 1      # inside, unaligned
        assert capi.lib.dvbs2_host_is_page_locked(base + 9 * page, 8 * page) == 1
        assert capi.lib.dvbs2_host_is_page_locked(base, 8 * page + 1) == 0              # runs past the registration
        assert capi.lib.dvbs2_host_is_page_locked(base + 4 * page, 10 * page) == 0      # spans the hole: both ends are page-locked
        assert capi.lib.dvbs2_host_is_page_locked(base + 8 * page, page) == 0           # the hole itself
    finally:
        capi.check(capi.lib.dvbs2_host_unregister(base))
        capi.check(capi.lib.dvbs2_host_unregister(base + 9 * page))
    assert capi.lib.dvbs2_host_is_page_locked(base, 8 * page) == 0
    t = torch.empty(1 << 20, dtype=torch.uint8).pin_memory()                            # hipHostMalloc'ed by the runtime
    assert capi.lib.dvbs2_host_is_page_locked(t.data_ptr(), t.numel()) == 1
    assert capi.lib.dvbs2_host_is_page_locked(t.data_ptr() + 5, t.numel() - 5) == 1
    assert capi.lib.dvbs2_host_is_page_locked(0, 16) == 0 and capi.lib.dvbs2_host_is_page_locked(base, 0) == 0
    del t


def test_host_entry_large_odd_group_with_registered_buffers():
    """page-locked caller buffers + a call of more than 1024 frames + a group size whose chunk unit (2 G for odd G) exceeds
    the first chunk of 512 frames: the chunk plan must stay inside the call (it used to place its first boundary at `unit` > n_frames
    and copy past the caller's buffers). G = 513 > 64: host-driven group resolution. Against the device entry and, for the first group,
    the scalar restatement."""
    import torch
    table, G, cap = "S2_TABLE_C1", 513, 4
    nf = 1026
    N, K, _, _ = T.ldpc_info(table)
    llr, _ = T.llr_codeword_awgn(table, nf, 91, amp=5, sigma=5.0)
    llr[5] = T.llr_noise(1, N, 3)[0]
    guard = 4096                                           # canaries behind every caller buffer
    import mmap

    def shared(n, dtype, fill):
        # the caller's registered buffers are mappings of their own (MAP_SHARED | MAP_ANONYMOUS: whole pages that nothing else lives in, no
        # copy-on-write, no anonymous huge pages) -- registering numpy HEAP memory is what faulted intermittently in bench.py (include/dvbs2_fec_hip.h)
        a = np.frombuffer(mmap.mmap(-1, n * np.dtype(dtype).itemsize), dtype=dtype)
        a[...] = fill
        return a
    xin = shared(llr.size + guard, np.int8, 0x55); xin[:llr.size] = llr.ravel()
    bits = shared(nf * (K // 8) + guard, np.uint8, 0xA5)
    ret = shared(2 + guard // 4, np.int32, 0x7A7A7A7A)
    for a in (xin, bits, ret):
        capi.check(capi.lib.dvbs2_host_register(a.ctypes.data, a.nbytes))
    dec = LdpcDecoder(table=table, message_bits=K, group_size=G, max_frames=2 * nf, max_trials=cap, outputmode=capi.OM_MESSAGE)
    try:
        capi.check(capi.lib.dvbs2_ldpc_decode(dec._h, xin.ctypes.data, nf, cap, capi.OM_MESSAGE, bits.ctypes.data, None, ret.ctypes.data))
    finally:
        for a in (xin, bits, ret):
            capi.check(capi.lib.dvbs2_host_unregister(a.ctypes.data))
    assert (bits[nf * (K // 8):] == 0xA5).all() and (ret[2:] == 0x7A7A7A7A).all() and (xin[llr.size:] == 0x55).all()
    d_in = torch.from_numpy(llr).cuda()
    d_bits = torch.empty((nf, K // 8), dtype=torch.uint8, device="cuda")
    d_ret = torch.empty(2, dtype=torch.int32, device="cuda")
    dec.work_device(d_in.data_ptr(), nf, d_bits.data_ptr(), 0, d_ret.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert np.array_equal(bits[:nf * (K // 8)].reshape(nf, K // 8), d_bits.cpu().numpy()) and ret[:2].tolist() == d_ret.cpu().tolist()
    want, wret = T.oracle_ldpc_decode(table, llr[:G], G, cap)
    assert ret[0] == wret[0] and np.array_equal(bits[:G * (K // 8)].reshape(G, K // 8), T.pack_bits(want, K))
    dec.close()


def test_host_entry_with_driver_allocated_buffers():
    """dvbs2_host_alloc / dvbs2_host_free (hipHostMalloc'ed caller buffers, what include/dvbs2_fec_hip.h recommends over registering malloc'ed
    memory): the range is recognised as page-locked, the host-pointer entry lands its results in it directly and they equal the device entry's."""
    import torch
    from dvbs2rx_amd import HostBuffer
    table, G, cap, nf = "S2_TABLE_C1", 32, 6, 640
    N, K, _, _ = T.ldpc_info(table)
    llr, _ = T.llr_codeword_awgn(table, nf, 93, amp=5, sigma=5.2)
    hx, hb, hr = HostBuffer((nf, N), np.int8), HostBuffer((nf, K // 8), np.uint8), HostBuffer((nf // G,), np.int32)
    assert hx.ptr % 4096 == 0
    for h in (hx, hb, hr):
        assert capi.lib.dvbs2_host_is_page_locked(h.ptr, h.nbytes) == 1
    assert capi.lib.dvbs2_host_is_page_locked(hb.ptr, hb.nbytes + (1 << 30)) == 0  # past the end of the allocation
    hx.array[...] = llr; hb.array[...] = 0xA5; hr.array[...] = 0x7A7A7A7A
    dec = LdpcDecoder(table=table, message_bits=K, group_size=G, max_frames=nf, max_trials=cap, outputmode=capi.OM_MESSAGE)
    capi.check(capi.lib.dvbs2_ldpc_decode(dec._h, hx.ptr, nf, cap, capi.OM_MESSAGE, hb.ptr, None, hr.ptr))
    d_in = torch.from_numpy(llr).cuda()
    d_bits = torch.empty((nf, K // 8), dtype=torch.uint8, device="cuda")
    d_ret = torch.empty(nf // G, dtype=torch.int32, device="cuda")
    dec.work_device(d_in.data_ptr(), nf, d_bits.data_ptr(), 0, d_ret.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert np.array_equal(hb.array, d_bits.cpu().numpy()) and hr.array.tolist() == d_ret.cpu().tolist()
    want, wret = T.oracle_ldpc_decode(table, llr[:G], G, cap)
    assert hr.array[0] == wret[0] and np.array_equal(hb.array[:G], T.pack_bits(want, K))
    dec.close()
    for h in (hx, hb, hr):
        h.free()
    assert capi.lib.dvbs2_host_free(None) == capi.OK
    p = C.c_void_p()
    assert capi.lib.dvbs2_host_alloc(C.byref(p), 0) == capi.EINVAL and capi.lib.dvbs2_host_alloc(None, 16) == capi.EINVAL


def test_misaligned_device_pointers_are_refused():
    """include/dvbs2_fec_hip.h: d_llr_in / d_llr_out of the *_device entry points are moved with 8-byte loads and stores; a misaligned
    pointer is DVBS2_EINVAL, not a GPU memory fault."""
    import torch
    table = "S2_TABLE_C1"
    N, K, _, _ = T.ldpc_info(table)
    dec = LdpcDecoder(table=table, message_bits=K, group_size=32, max_frames=32, max_trials=5, outputmode=capi.OM_MESSAGE)
    d_in = torch.zeros(32 * N + 16, dtype=torch.int8, device="cuda")
    d_out = torch.zeros(32 * N + 16, dtype=torch.int8, device="cuda")
    d_bits = torch.zeros((32, K // 8), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    f = capi.lib.dvbs2_ldpc_decode_device
    assert f(dec._h, d_in.data_ptr() + 4, 32, 5, capi.OM_MESSAGE, d_bits.data_ptr(), None, None, st) == capi.EINVAL
    assert f(dec._h, d_in.data_ptr(), 32, 5, capi.OM_MESSAGE, d_bits.data_ptr(), d_out.data_ptr() + 1, None, st) == capi.EINVAL
    assert capi.lib.dvbs2_ldpc_enqueue_device(dec._h, d_in.data_ptr() + 2, 32, 5, capi.OM_MESSAGE, d_bits.data_ptr(), None, None, st) == capi.EINVAL
    capi.check(f(dec._h, d_in.data_ptr() + 8, 32, 5, capi.OM_MESSAGE, d_bits.data_ptr(), d_out.data_ptr() + 8, None, st))  # 8-byte offsets are fine
    torch.cuda.synchronize()
    dec.close()
