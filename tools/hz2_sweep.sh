#!/bin/bash
# tools/hz2_sweep.sh -- GPU box: every table whose degree class has the heavy-hazard build, policy build vs DVBS2_HZ2=1, frames/s
# (4096 frames, 50 updates, never-converging input). The tables that gain go into csrc/ldpc_policy_hz2.inc.
python - <<'PY'
import os, sys, subprocess
ROOT = os.path.dirname(os.path.abspath('tools'))
sys.path.insert(0, 'gr-dvbs2rx_amd/python')
from dvbs2rx_amd import ldpc_table_names, ldpc_table_info
for t in ldpc_table_names():
    i = ldpc_table_info(t)
    if i['conflict_layers'] == 0: continue
    nf = 4096 if i['N'] == 64800 else 8192
    r = []
    for hz in ('0', '1'):
        env = dict(os.environ, DVBS2_HZ2=hz, WARM_S='0.3')
        out = subprocess.run([sys.executable, 'tools/exp_tables.py', f'{t}:50:{nf}'], env=env, capture_output=True, text=True).stdout
        fr = [l for l in out.splitlines() if 'fr/s' in l]
        r.append(float(fr[0].split('ms')[1].split('fr/s')[0]) if fr else 0.0)
    print(f'{t:16s} base {r[0]:9.0f} hz2 {r[1]:9.0f}  {r[1]/max(r[0],1):5.2f}', flush=True)
PY
