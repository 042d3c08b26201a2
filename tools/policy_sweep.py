#!/usr/bin/env python3
"""GPU box: frames/s of every table under each sweep-kernel build (packed nodes on/off x one-frame workgroups on/off; the
parity-in-records and dense choices stay with the library). Output: one line per table with the four rates and the best."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-dvbs2rx_amd", "python"))
from dvbs2rx_amd import ldpc_table_names, ldpc_table_info, ldpc_layer_info
tables = sys.argv[1:] or ldpc_table_names()
for t in tables:
    i = ldpc_table_info(t)
    degs = [ldpc_layer_info(t, l)["cnt"] + 2 for l in range(i["q"])]
    nf, cap = (4096, 50) if i["N"] == 64800 else (16384, 25)
    res = {}
    for v2 in (1, 0):
        for solo in (1, 0):
            env = dict(os.environ, DVBS2_V2=str(v2), DVBS2_CHAIN_V2=str(v2), DVBS2_SOLO=str(solo), WARM_S="0.2")
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "exp_tables.py"), f"{t}:{cap}:{nf}"], capture_output=True, text=True, env=env).stdout
            w = out.strip().split("\n")[-1].split()
            res[(v2, solo)] = float(w[w.index("fr/s") - 1]) if "fr/s" in w else 0.0
    best = max(res, key=res.get)
    print(f"{t:14s} deg {min(degs):2d}-{max(degs):2d} q {i['q']:3d} haz {i['conflict_layers']:2d} | packed+solo {res[(1,1)]:9.0f} packed {res[(1,0)]:9.0f} plain+solo {res[(0,1)]:9.0f} plain {res[(0,0)]:9.0f} | best packed={best[0]} solo={best[1]}", flush=True)
