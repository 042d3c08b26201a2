#!/bin/bash
# round 5, call 2: the tree after the status-word group stop + ADVICE fixes (tests, bench), the corrected one-word bound, cycle stamps
O=gpurun_out/r5b; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -5 $O/pytest_gpu.log
timeout 600 python bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err; echo "bench rc $?"; tail -c 600 $O/bench.err
python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/r5b/bench.json') if l.startswith('{')][-1])
print('headline', round(d['value']), 'frac', round(d['roofline']['frac'],4), 'limiter', d['roofline'].get('limiter'))
for k,c in d.get('configs',{}).items():
    print(k, round(c.get('value',0)), c.get('roofline',{}).get('frac'), c.get('mean_updates_per_group'), c.get('frac_of_proportional_rate'), str(c.get('parity'))[:60])
print(json.dumps(d['configs'].get('config2_host',{}).get('pipelined'))[:1200])
print(json.dumps(d['configs'].get('config3_awgn',{}))[:1500])
PY
python tools/abx.py --out $O/oneword.txt --spec tree --spec "ow=libdvbs2_fec_hip_oneword.so" \
  S2_TABLE_B4:50:4096 S2X_TABLE_B3:50:4096 S2_TABLE_B7:50:4096 S2X_TABLE_B8:50:4096 S2X_TABLE_B21:50:4096 S2_TABLE_B11:50:4096
for t in S2_TABLE_B11 S2_TABLE_B7 S2_TABLE_B9; do
  DVBS2_TIMING=1 DVBS2_TIMING_LAYERS=1 DVBS2_LIB=$PWD/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip_timing.so timeout 300 python tools/exp_tables.py $t:50:1024 > $O/stamps_$t.txt 2>&1
  tail -70 $O/stamps_$t.txt | grep -E "layer|hazard|timing" | tail -60
done
