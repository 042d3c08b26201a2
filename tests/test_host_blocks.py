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


def test_demapper_block_stream(exe, tmp_path):
    rng = np.random.default_rng(9)
    syms = (np.exp(1j * (rng.integers(0, 4, (2, 8100)) * np.pi / 2 + np.pi / 4)) +
            0.2 * (rng.normal(size=(2, 8100)) + 1j * rng.normal(size=(2, 8100)))).astype(np.complex64)
    out, log = run(exe, tmp_path, "demap", syms, capi.FECFRAME_SHORT, "C1_4", capi.MOD_QPSK)
    want = T.oracle_demap(syms, np.float32(1.0) / np.float32(10.0), 4)
    assert np.array_equal(out.view(np.int8).reshape(2, -1), want)
    assert "snr_db 10.000" in log
