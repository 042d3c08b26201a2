#!/bin/bash
# round 5, call 9: whole -m gpu suite + bench + fuzz soak on the tree with the re-decided policy (B9-B11 packed, class-32 pure build)
O=gpurun_out/r5i; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_gpu.log
timeout 600 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc $?"
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5i/bench.json') if l.startswith('{')][-1])
print('headline', round(d['value']), 'frac', round(d['roofline']['frac'],4))
for k,c in d.get('configs',{}).items():
    print(k, round(c.get('value',0)), c.get('roofline',{}).get('frac'), c.get('roofline',{}).get('kernel'), c.get('mean_updates_per_group'), c.get('frac_of_proportional_rate'))
PY
FUZZ_S=240 bash tools/r5/fuzz.sh r5i
