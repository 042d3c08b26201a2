#!/bin/bash
# round 5, call 3: status report as atomic max (tree) vs plain store (ststore) -- short 1/4 lost 40 % with the store
O=gpurun_out/r5c; mkdir -p $O
python tools/abx.py --out $O/status.txt --spec tree --spec "store=libdvbs2_fec_hip_ststore.so" S2_TABLE_C1:25:16384 S2_TABLE_C3:25:16384 S2_TABLE_B4:50:4096 S2X_TABLE_C8:25:16384 S2_TABLE_B1:50:4096
timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --only config4,config2_awgn,config4_awgn > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5c/bench.json') if l.startswith('{')][-1])
print('headline', round(d['value']), 'frac', round(d['roofline']['frac'],4))
for k,c in d.get('configs',{}).items():
    print(k, round(c.get('value',0)), c.get('roofline',{}).get('frac'), c.get('mean_updates_per_group'), c.get('frac_of_proportional_rate'), (c.get('pipelined') or {}).get('frac_of_proportional_rate'))
PY
