#!/bin/bash
# GPU box, round 6 call 9: parity-in-records kernel with buffer-descriptor record accesses -- bit-exactness and A/B
O=gpurun_out/r6i; mkdir -p $O
timeout 1200 python -m pytest tests/test_ldpc_gpu.py -x -q -n 4 -k "policy or pr-byte or C1 or C4 or group or near or saturation or counters or config1 or baseline" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
timeout 2400 python tools/abx.py --out $O/ab.txt --reps 3 --spec "base=libdvbs2_fec_hip_base.so" --spec tree \
  S2_TABLE_C1:25:16384 S2_TABLE_C2:25:16384 S2_TABLE_C3:25:16384 S2_TABLE_C4:25:16384 S2X_TABLE_C1:25:16384 S2X_TABLE_C2:25:16384 S2X_TABLE_C3:25:16384 S2X_TABLE_C8:25:8192 S2X_TABLE_C9:25:8192 S2X_TABLE_C10:25:8192
for L in base ""; do
  DVBS2_LIB=$PWD/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip${L:+_$L}.so python tools/exp_awgn2.py 16384 0.5 7 S2_TABLE_C1 25 2>&1 | tail -1 | sed "s/^/[${L:-tree}] /"
done > $O/awgn.txt 2>&1; cat $O/awgn.txt
