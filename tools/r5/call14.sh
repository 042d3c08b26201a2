#!/bin/bash
# round 5, call 14: rocprofv3 passes of the round (tools/profile_round.sh r05), a driver-style bench run, every table
bash tools/profile_round.sh r05 > /dev/null 2>&1; tail -30 gpurun_out/r05/summary.txt | cut -c1-250
O=gpurun_out/r05full; mkdir -p $O
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r05full/bench.json') if l.startswith('{')][-1])
print('headline', round(d['value']), 'frac', round(d['roofline']['frac'],4), d['roofline'].get('limiter'))
for k,c in d.get('configs',{}).items():
    print(k, round(c.get('value',0)), c.get('roofline',{}).get('frac'), c.get('ms_per_step_runs'), c.get('mean_updates_per_group'), c.get('frac_of_proportional_rate'))
PY
timeout 1500 python tools/all_tables_perf.py > $O/all_tables.md 2> $O/all_tables.err; tail -3 $O/all_tables.md
