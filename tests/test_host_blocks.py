"""GPU: the C++ host-side mirror of the reference blocks (make / forecast / general_work / getters) over the C ABI."""
import os
import subprocess

import numpy as np
import pytest

import fec_testlib as T
from dvbs2rx_amd import capi, get_fec_info

pytestmark = pytest.mark.gpu
LIBDIR = os.path.join(T.ROOT, "gr-dvbs2rx_amd", "lib")


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    out = str(tmp_path_factory.mktemp("host") / "host_blocks_main")
    subprocess.check_call(["g++", "-O2", "-std=c++17", os.path.join(T.ROOT, "tests", "host_blocks_main.cpp"), "-o", out,
                           "-L" + LIBDIR, "-ldvbs2_fec_hip", "-Wl,-rpath," + LIBDIR])
    return out


def run(exe, tmp_path, kind, data, framesize, rate, arg):
    fin, fout = str(tmp_path / "in.bin"), str(tmp_path / "out.bin")
    data.tofile(fin)
    r = subprocess.run([exe, kind, fin, fout, str(framesize), rate, str(arg)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    return np.fromfile(fout, np.uint8), r.stdout


def test_ldpc_block_stream(exe, tmp_path):
    table = "S2_TABLE_C1"
    N, K, _, _ = T.ldpc_info(table)
    llr, _ = T.llr_codeword_awgn(table, 64, 41, amp=5, sigma=6.0)
    out, log = run(exe, tmp_path, "ldpc", llr, capi.FECFRAME_SHORT, "C1_4", 25)
    want, ret = T.oracle_ldpc_decode(table, llr, 32, 25)
    assert np.array_equal(out.reshape(64, K // 8), T.pack_bits(want, K))
    avg = sum(25 if r < 0 else 25 - r for r in ret) // 2
    assert f"avg_trials {avg} pdu_frames 64" in log and f"consumed {64 * N} produced {64 * K // 8}" in log


def test_bch_block_stream(exe, tmp_path):
    fi = get_fec_info(capi.STANDARD_DVBS2, capi.FECFRAME_SHORT, "C1_4")
    m, prim = T.BCH_FIELDS[capi.FECFRAME_SHORT]
    ob = T.OracleBch(m, prim, fi["bch_t"], fi["bch_n"])
    rng = np.random.default_rng(3)
    msg = rng.integers(0, 256, (8, ob.k // 8), dtype=np.uint8)
    cw = ob.encode_bytes(msg)
    rx = np.stack([T.flip_bits(cw[i], rng.choice(ob.n, [0, 1, 5, 12, 12, 3, 30, 2][i], replace=False)) for i in range(8)])
    want, wret = ob.decode_bytes(rx)
    if (wret == -2).any():
        pytest.skip("pattern hits a reference-throws case")
    out, log = run(exe, tmp_path, "bch", rx, capi.FECFRAME_SHORT, "C1_4", 8)
    assert np.array_equal(out.reshape(8, -1), want)
    assert f"frames 8 errors {int((wret == -1).sum())}" in log


def test_bbdeheader_block_stream(exe, tmp_path):
    """bbdeheader_bb through the C++ mirror, in two general_work calls: TS bytes and the five counters of the block."""
    from test_bbdeheader import _fuzz_stream
    fi = get_fec_info(capi.STANDARD_DVBS2, capi.FECFRAME_NORMAL, "C1_4")
    fr = _fuzz_stream(fi["bch_k"], 40, 55)
    orc = T.OracleBbDeheader(fi["bch_k"])
    want = orc.work(fr)
    c = orc.counters()
    nf = fr.shape[0]  # (the stream generator loses a few BBFRAMEs on purpose)
    out, log = run(exe, tmp_path, "bbdh", fr, capi.FECFRAME_NORMAL, "C1_4", nf)
    assert out.size == want.size and np.array_equal(out, want), (out.size, want.size, log)
    assert f"packets {c['packets']} errors {c['errors']} bbframes {nf} dropped {c['dropped']} gaps {c['gaps']}" in log
    assert f"consumed {fr.size} produced {want.size}" in log


def test_demapper_block_stream(exe, tmp_path):
    rng = np.random.default_rng(9)
    syms = (np.exp(1j * (rng.integers(0, 4, (2, 8100)) * np.pi / 2 + np.pi / 4)) +
            0.2 * (rng.normal(size=(2, 8100)) + 1j * rng.normal(size=(2, 8100)))).astype(np.complex64)
    out, log = run(exe, tmp_path, "demap", syms, capi.FECFRAME_SHORT, "C1_4", capi.MOD_QPSK)
    want = T.oracle_demap(syms, np.float32(1.0) / np.float32(10.0), 4)
    assert np.array_equal(out.view(np.int8).reshape(2, -1), want)
    assert "snr_db 10.000" in log


def test_demapper_llr_pdu_feedback_loop(exe, tmp_path):
    """demapper -> LDPC -> llr_pdu -> demapper.handle_llr_pdu (reference :188-318): the first 32 frames use the
    per-frame pre-decoder N0, the refinement averages the per-frame post-decoder estimates, frames 32..63 use it."""
    table = "S2_TABLE_C1"
    N, K, _, _ = T.ldpc_info(table)
    rng = np.random.default_rng(17)
    cw = T.ldpc_encode(table, rng.integers(0, 2, (64, K), dtype=np.uint8))
    pts = ((1 - 2.0 * cw[:, 0::2]) + 1j * (1 - 2.0 * cw[:, 1::2])) * np.sqrt(0.5)
    syms = (pts + 0.55 * (rng.normal(size=pts.shape) + 1j * rng.normal(size=pts.shape))).astype(np.complex64)
    out, log = run(exe, tmp_path, "loop", syms, capi.FECFRAME_SHORT, "C1_4", capi.MOD_QPSK)
    out = out.view(np.int8).reshape(64, N)
    o = T.oracle()
    pre = np.array([o.oracle_demap_snr(T.ptr(np.ascontiguousarray(syms[f])), N // 2, 4) for f in range(32)], np.float32)
    llr0 = T.oracle_demap(syms[:32], np.float32(1.0) / pre, 4)
    # N0 is a float reduction (tolerance 2e-4): allow the rare LLR that sits on a rounding boundary
    assert np.mean(out[:32] != llr0) < 1e-3 and np.abs(out[:32].astype(int) - llr0).max() <= 1
    dec, _ = T.oracle_ldpc_decode(table, out[:32], 32, 25)
    ref = np.float32(np.mean([o.oracle_demap_snr_refined(T.ptr(np.ascontiguousarray(syms[f])), T.ptr(np.ascontiguousarray(dec[f])),
                                                         N // 2, 4, 0) for f in range(32)]))
    got = float(log.split("snr_lin")[1].split()[0])
    assert "found 32" in log and abs(got - ref) / ref < 5e-4
    assert abs(ref - 1 / (2 * 0.55 ** 2)) / ref < 0.05  # decoded frames => the true SNR
    llr1 = T.oracle_demap(syms[32:], np.float32(1.0) / np.float32(got), 4)
    assert np.mean(out[32:] != llr1) < 1e-3 and np.abs(out[32:].astype(int) - llr1).max() <= 1


def test_three_blocks_equal_the_fused_host_entry(exe, tmp_path):
    """C++ only, through the C ABI: xfecframe_demapper_cb -> ldpc_decoder_bb -> bch_decoder_bb (the mirror's general_work on host
    buffers, one block after the other) give the bytes of ONE dvbs2_chain_decode call on the same symbols; both equal the CPU chain."""
    framesize, rate, nf = capi.FECFRAME_SHORT, "C1_2", 96  # (whole groups of 32: the LDPC block's granule)
    fi = get_fec_info(capi.STANDARD_DVBS2, framesize, rate)
    m, prim = T.BCH_FIELDS[framesize]
    ob = T.OracleBch(m, prim, fi["bch_t"], fi["bch_n"])
    rng = np.random.default_rng(77)
    sent = rng.integers(0, 256, (nf, fi["bch_k"] // 8), dtype=np.uint8)
    cw = T.ldpc_encode(fi["table"], np.unpackbits(ob.encode_bytes(sent), axis=1))
    pts = ((1 - 2.0 * cw[:, 0::2]) + 1j * (1 - 2.0 * cw[:, 1::2])) * np.sqrt(0.5)
    syms = (pts + np.sqrt(1 / 14.0) * (rng.normal(size=pts.shape) + 1j * rng.normal(size=pts.shape))).astype(np.complex64)  # Es/N0 = 7
    syms[11] *= 0.05  # one frame below the noise
    out, log = run(exe, tmp_path, "chain", syms, framesize, rate, capi.MOD_QPSK)
    llr = T.oracle_demap(syms, np.float32(1.0) / np.float32(7.0), 4)
    dec = np.concatenate([T.oracle_ldpc_decode(fi["table"], llr[a:a + 32], 32, 25)[0] for a in range(0, nf, 32)])
    want, wcorr = ob.decode_bytes(T.pack_bits(dec, fi["bch_n"]))
    assert f"frames {nf} fused_equals_blocks 1" in log, log
    assert f"bch_failures {int((wcorr < 0).sum())} block_errors {int((wcorr < 0).sum())}" in log, log
    assert np.array_equal(out.reshape(nf, -1), want)
    good = wcorr >= 0
    assert good.sum() >= nf - 2 and np.array_equal(want[good], sent[good])
