#!/bin/bash
# GPU box, round-5 checkpoint: whole -m gpu suite, smoke, fuzz soak, rocprofv3 passes (tools/profile_round.sh), a driver-style bench run, every table
TAG=${1:-r05}; O=gpurun_out/${TAG}full; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
FUZZ_S=${FUZZ_S:-200} bash tools/r5/fuzz.sh ${TAG}full
bash tools/profile_round.sh $TAG > /dev/null 2>&1
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - "$O" <<'PY'
import json, sys
d=json.loads([l for l in open(sys.argv[1]+'/bench.json') if l.startswith('{')][-1])
print('headline', round(d['value']), 'frac', round(d['roofline']['frac'],4), (d['roofline'].get('limiter') or {}).get('verdict'))
for k,c in d.get('configs',{}).items():
    print(k, round(c.get('value',0)), c.get('roofline',{}).get('frac'), c.get('ms_per_step_runs'), c.get('mean_updates_per_group'), c.get('frac_of_proportional_rate'))
PY
timeout 1500 python tools/all_tables_perf.py > $O/all_tables.md 2> $O/all_tables.err; tail -2 $O/all_tables.md
