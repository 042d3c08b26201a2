#!/usr/bin/env python3
"""GPU box: frames/s of every built-in LDPC table (noise input, every frame runs the full cap): normal frames 4096 x 50
updates, short/medium 16384 x 25. Prints a markdown table."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gr-dvbs2rx_amd", "python"))
from dvbs2rx_amd import ldpc_table_names, ldpc_table_info, ldpc_layer_info, LdpcDecoder
print("| table | N | K | q | check degree | hazard layers | kernel | frames/s | coded Gbit/s |")
print("|---|---|---|---|---|---|---|---|---|")
for t in ldpc_table_names():
    i = ldpc_table_info(t)
    degs = [ldpc_layer_info(t, l)["cnt"] + 2 for l in range(i["q"])]
    nf, cap = (4096, 50) if i["N"] == 64800 else (16384, 25)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "exp_tables.py"), f"{t}:{cap}:{nf}"], capture_output=True, text=True).stdout
    fps = [w for w in out.strip().split("\n")[-1].split()]
    fps = float(fps[fps.index("fr/s") - 1])
    d = LdpcDecoder(table=t, message_bits=i["K"], group_size=32, max_frames=32, max_trials=5)
    kn = d.kernel_name; d.close()
    dd = f"{min(degs)}" if min(degs) == max(degs) else f"{min(degs)}-{max(degs)}"
    print(f"| {t} | {i['N']} | {i['K']} | {i['q']} | {dd} | {i['conflict_layers']} | {kn} | {fps/1e3:.1f} k | {fps*i['N']/1e9:.2f} |", flush=True)
