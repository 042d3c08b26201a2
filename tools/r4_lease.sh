#!/bin/bash
# tools/r4_lease.sh <tag> -- one lease of the round-4 robustness series: the five BASELINE kernels, config 5 with the
# two-level lane chain on / off, and the policy sweep of all 57 tables, all on ONE box; run on >= 2 leases and compare.
TAG=$1; O=gpurun_out/r4lease_$TAG; mkdir -p $O
hostname > $O/box.txt; rocm-smi --showserial --showbus 2>/dev/null | grep -E "Serial|Bus" >> $O/box.txt
python bench.py --no-cpu-baseline --gate first --steps 5 > $O/bench.json 2> $O/bench.err
for rep in 1 2; do for tl in 1 0; do
  DVBS2_TWO_LEVEL=$tl python bench.py --no-cpu-baseline --gate first --only config5 --steps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['configs']['config5']
print('two_level=$tl rep=$rep config5', round(c['value']), 'fr/s  launch', round(c['roofline']['avg_launch_ms'],2), 'ms  headline', round(d['value']))" >> $O/config5_tl.log
done; done
cat $O/config5_tl.log
python tools/policy_sweep.py > $O/policy_sweep.log 2>&1
tail -3 $O/policy_sweep.log
