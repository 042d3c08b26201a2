#!/bin/bash
# GPU box, round 6 call 8: the syndrome phase with fewer barriers -- bit-exactness over every build, operating-point tests, A/B against the previous library
O=gpurun_out/r6h; mkdir -p $O
timeout 1500 python -m pytest tests/test_ldpc_gpu.py -x -q -n 4 > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
timeout 3000 python tools/abx.py --out $O/ab.txt --reps 3 --spec "base=libdvbs2_fec_hip_base.so" --spec tree \
  S2_TABLE_B4:50:4096 S2_TABLE_B7:50:4096 S2_TABLE_B11:50:4096 S2X_TABLE_B21:50:4096 S2_TABLE_C1:25:16384 S2_TABLE_B9:50:4096 S2_TABLE_B10:50:4096 S2_TABLE_B8:50:4096 S2_TABLE_B5:50:4096 \
  S2X_TABLE_B10:50:4096 S2_TABLE_B1:50:4096 S2_TABLE_B2:50:4096 S2_TABLE_C7:25:8192 S2_TABLE_C5:25:8192 S2_TABLE_C3:25:16384 S2_TABLE_C10:25:8192 S2X_TABLE_B3:50:4096 S2X_TABLE_B16:50:4096 S2X_TABLE_C8:25:8192 T2_TABLE_B3:25:8192
for a in "S2_TABLE_B4 2.0 50 4096" "S2_TABLE_C1 0.5 25 16384"; do set -- $a
  for L in base ""; do
    DVBS2_LIB=$PWD/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip${L:+_$L}.so python tools/exp_awgn2.py $4 $2 7 $1 $3 2>&1 | tail -1 | sed "s/^/[${L:-tree}] /"
  done
done > $O/awgn.txt 2>&1; cat $O/awgn.txt
timeout 600 python -m pytest tests/test_shard_gloo.py tests/test_bch_demap_gpu.py -x -q -m gpu -k "rccl or chain_host" > $O/pytest2.log 2>&1; echo "pytest2 rc $?"; tail -2 $O/pytest2.log
timeout 900 python bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline --gate none --only config3_host > $O/bench_host.json 2> $O/bench_host.err
python - "$O" <<'PY'
import json, sys
d=json.loads([l for l in open(sys.argv[1]+'/bench_host.json') if l.startswith('{')][-1])
h=d['configs']['config3_host']
for k in ('worst_case','operating_point'):
    for n,v in h[k]['calls'].items(): print(k,n, round(v['frames_per_s']), round(v['frac_of_resident'],3), round(v['frac_of_link_bound'],3))
    for n,v in ((a,b) for a,b in h[k]['pipelined'].items() if isinstance(b,dict)): print(k,'pipe',n, round(v['frames_per_s']), round(v['frac_of_resident'],3), round(v['frac_of_link_bound'],3))
PY
