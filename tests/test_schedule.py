"""Host logic: the (group, shift) schedule compiler vs the reference's pos[] construction (as restated
and pinned in oracle/ldpc_oracle.c) and the hazard data of SURVEY.md Appendix A/B."""
import ctypes as C

import numpy as np
import pytest

import fec_testlib as T
from dvbs2rx_amd import ldpc_layer_info, ldpc_table_info

# (N, K, q, LINKS_TOTAL, conflict layers)
EXPECT = {
    "S2_TABLE_B4": (64800, 32400, 90, 226799, 8),
    "S2_TABLE_B7": (64800, 48600, 45, 226799, 20),
    "S2_TABLE_B11": (64800, 58320, 18, 194399, 18),
    "S2_TABLE_C1": (16200, 3240, 36, 48599, 4),
    "S2X_TABLE_B21": (64800, 55440, 26, 273239, 0),
    "S2X_TABLE_C8": (32400, 6480, 72, 103679, 0),
    "T2_TABLE_A3": (64800, 43200, 60, 215999, 11),
}


@pytest.mark.parametrize("table", sorted(EXPECT))
def test_table_info(table):
    i = ldpc_table_info(table)
    assert (i["N"], i["K"], i["q"], i["links_total"], i["conflict_layers"]) == EXPECT[table]
    assert T.ldpc_info(table) == EXPECT[table][:4]


def test_conflict_blocks_b4():
    # B4 conflict layers i:B_i
    want = {1: 11, 57: 171, 58: 41, 62: 38, 63: 47, 71: 164, 74: 118, 86: 36}
    got = {}
    for i in range(90):
        b = ldpc_layer_info("S2_TABLE_B4", i)["block"]
        if b != 360:
            got[i] = b
    assert got == want


def test_conflict_blocks_b7_b11():
    assert ldpc_layer_info("S2_TABLE_B7", 18)["block"] == 2
    assert ldpc_layer_info("S2_TABLE_B11", 5)["block"] == 4
    assert all(ldpc_layer_info("S2X_TABLE_B21", i)["block"] == 360 for i in range(26))


@pytest.mark.parametrize("table", ["S2_TABLE_C1", "S2_TABLE_B4", "S2X_TABLE_C4"])
def test_entries_reproduce_encoder(table):
    """Check (i, j) must read data bit 360*g + (j - s) mod 360: rebuild H from the schedule and verify
    that codewords of the (independently written) IRA encoder satisfy every check."""
    info = ldpc_table_info(table)
    N, K, q = info["N"], info["K"], info["q"]
    rng = np.random.default_rng(7)
    cw = T.ldpc_encode(table, rng.integers(0, 2, (2, K), dtype=np.uint8))
    j = np.arange(360)
    for f in range(2):
        par = cw[f, K:]
        for i in range(q):
            li = ldpc_layer_info(table, i)
            acc = par[q * j + i].astype(np.int64)
            if i:
                acc = acc + par[q * j + i - 1]
            else:
                prev = np.zeros(360, np.int64)
                prev[1:] = par[q * (j[1:] - 1) + q - 1]
                acc = acc + prev
            for g, s in zip(li["groups"], li["shifts"]):
                acc = acc + cw[f, 360 * g + (j - s) % 360]
            assert not (acc & 1).any(), (table, i)
