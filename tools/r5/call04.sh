#!/bin/bash
# round 5, call 4: config 4 through bench.py with the status report as a plain store (ststore) and as an atomic max (tree), alternately
O=gpurun_out/r5d; mkdir -p $O
for rep in 1 2; do for lib in tree ststore; do
  e=""; [ $lib != tree ] && e="DVBS2_LIB=$PWD/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip_$lib.so"
  env $e timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --gate first --only config4 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); c=d['configs']['config4']
print('$lib rep $rep config4', round(c['value']), 'launch ms', round(c['roofline']['avg_launch_ms'],2), 'headline', round(d['value']))" | tee -a $O/config4.txt
done; done
