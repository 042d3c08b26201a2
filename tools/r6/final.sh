#!/bin/bash
# GPU box, round-6 checkpoint: whole -m gpu suite, smoke, a driver-style bench run, rocprofv3 passes (tools/profile_round.sh), fuzz soak, every table
TAG=${1:-r06}; O=gpurun_out/${TAG}full; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -3 $O/bench.err
python - "$O" <<'PY'
import json, sys
d=json.loads([l for l in open(sys.argv[1]+'/bench.json') if l.startswith('{')][-1])
print('headline', round(d['value']), 'frac', round(d['roofline']['frac'],4), (d['roofline'].get('limiter') or {}).get('verdict'), d.get('shader_clock'))
for k,c in d.get('configs',{}).items():
    print(k, round(c.get('value',0)), (c.get('roofline') or {}).get('frac'), c.get('ms_per_step_spread'), c.get('mean_updates_per_group'), c.get('frac_of_proportional_rate'), (c.get('pipelined') or {}).get('frac_of_proportional_rate'))
print('demap', d.get('demap'))
PY
if [ "${SKIP_PROF:-0}" != "1" ]; then bash tools/profile_round.sh $TAG > /dev/null 2>&1; tail -5 gpurun_out/$TAG/summary.txt; fi
if [ "${FUZZ_S:-0}" != "0" ]; then FUZZ_S=$FUZZ_S bash tools/r5/fuzz.sh ${TAG}full; fi
if [ "${ALL_TABLES:-0}" = "1" ]; then timeout 1500 python tools/all_tables_perf.py > $O/all_tables.md 2> $O/all_tables.err; tail -2 $O/all_tables.md; fi
