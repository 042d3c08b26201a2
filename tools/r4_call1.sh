#!/bin/bash
# round 4, first GPU call: tests, default bench, sticky pre-test A/B at the operating point
O=gpurun_out/r4a; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; echo "pytest rc $?" >> $O/gpu_tests.log
tail -3 $O/gpu_tests.log
python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc $?"
for s in 1 0 1 0; do
  DVBS2_STICKY_PRETEST=$s python bench.py --no-cpu-baseline --only config2_awgn --gate first 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); a=d['configs']['config2_awgn']
print('sticky=$s headline', round(d['value']), 'awgn', round(a['value']), 'mean', a['mean_updates_per_group'], 'frac_prop', round(a['frac_of_proportional_rate'],4))" >> $O/sticky_ab.log
done
cat $O/sticky_ab.log
