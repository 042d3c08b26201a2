import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-dvbs2rx_amd", "python")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import fec_testlib as T
from dvbs2rx_amd import LdpcDecoder, capi, ldpc_layer_info, ldpc_table_info
table = sys.argv[1] if len(sys.argv) > 1 else "S2_TABLE_B4"
trials = int(sys.argv[2]) if len(sys.argv) > 2 else 1
N, K, q, _ = T.ldpc_info(table)
x = T.llr_noise(2, N, 4242)
def run():
    dec = LdpcDecoder(table=table, message_bits=K, group_size=1, max_frames=2, max_trials=trials, outputmode=capi.OM_CODEWORD)
    b, o, r = dec.work(x, want_llr=True); dec.close(); return o
want, _ = T.oracle_ldpc_decode(table, x, 1, trials)
got = run()
bad = np.nonzero(got[0] != want[0])[0]
print(table, "trials", trials, "mismatching LLRs in frame 0:", len(bad))
if len(bad):
    groups = sorted(set(int(b) // 360 for b in bad if b < K))
    print(" data groups hit:", groups[:40], " parity bits hit:", int((bad >= K).sum()))
    for l in range(q):
        li = ldpc_layer_info(table, l)
        if li["block"] < 360:
            gs = [g for g in set(li["groups"]) if li["groups"].count(g) > 1]
            print("  hazard layer", l, "block", li["block"], "pair group", gs, "hit" if any(g in groups for g in gs) else "")
    only = os.environ.get("DVBS2_CHAIN_ONLY")
    if only:
        li = ldpc_layer_info(table, int(only))
        rows = {}
        for b in bad:
            if b < K:
                g = int(b) // 360
                for gg, sh in zip(li["groups"], li["shifts"]):
                    if gg == g: rows.setdefault((g, sh), []).append((int(b) % 360 + sh) % 360)  # check row touching this bit via (g, sh)
        for k, v in sorted(rows.items()): print("   entry", k, "rows", sorted(v)[:24], "n", len(v))
        print("   layer groups/shifts", list(zip(li["groups"], li["shifts"])))
    print(" first bad indices:", bad[:20], "got", got[0][bad[:10]], "want", want[0][bad[:10]])
