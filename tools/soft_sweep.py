#!/usr/bin/env python3
"""GPU box: software frame barriers (DVBS2_SOFT_BARRIER=1) against the policy build on every table of the degree classes >= 20 that HAS
hazard layers (tables without them take the software barriers by rule). Interleaved, best of two. -> csrc/ldpc_policy_soft.inc by hand /
tools/gen_policy_soft.py.   usage: soft_sweep.py > gpurun_out/<dir>/soft_sweep.txt"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-dvbs2rx_amd", "python"))
from dvbs2rx_amd import ldpc_table_names, ldpc_table_info, ldpc_layer_info


def fps(table, spec, env):
    e = dict(os.environ); e.update(env)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "exp_tables.py"), f"{table}:{spec}"], env=e, capture_output=True, text=True).stdout
    w = out.split()
    return float(w[w.index("fr/s") - 1]) if "fr/s" in w else 0.0


for t in ldpc_table_names():
    ti = ldpc_table_info(t)
    if ti["conflict_layers"] == 0:
        continue
    dmax = max(ldpc_layer_info(t, i)["cnt"] + 2 for i in range(ti["q"]))
    if dmax <= 16:
        continue
    spec = "50:4096" if ti["N"] == 64800 else "25:16384"
    a = [fps(t, spec, {}), fps(t, spec, {"DVBS2_SOFT_BARRIER": "1"}), fps(t, spec, {}), fps(t, spec, {"DVBS2_SOFT_BARRIER": "1"})]
    base, soft = max(a[0], a[2]), max(a[1], a[3])
    print(f"{t:16s} N={ti['N']} q={ti['q']:3d} hazard layers {ti['conflict_layers']:2d} max degree {dmax:2d}: policy {base:9.0f}  soft {soft:9.0f}  {soft / base:6.3f}", flush=True)
