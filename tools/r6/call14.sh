#!/bin/bash
# GPU box, round 6 call 14: tree = direct-minimum degree-3 / 4 node; pre-test of the parity-in-records kernel with both record loads ahead
# (-DDVBS2_PR_PRETEST_AHEAD=1: one-dword records only, =2: both kernels) -- bit-exactness and A/B
O=gpurun_out/r6n; mkdir -p $O
SEL="policy or pr-byte or C1 or C2 or C3 or C4 or C8 or C9 or group or near or saturation or counters or config1 or baseline or (test_every_table_bit_exact and (S2_TABLE_B1- or S2X_TABLE_B1-))"
timeout 1200 python -m pytest tests/test_ldpc_gpu.py -q -n 4 -k "$SEL" > $O/pytest_tree.log 2>&1; echo "pytest tree rc $?"; tail -2 $O/pytest_tree.log
for v in 1 2; do
DVBS2_LIB=$PWD/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip_ahead$v.so timeout 1200 python -m pytest tests/test_ldpc_gpu.py -q -n 4 -k "$SEL" > $O/pytest_ahead$v.log 2>&1; echo "pytest ahead$v rc $?"; tail -2 $O/pytest_ahead$v.log
done
timeout 1800 python tools/abx.py --out $O/ab.txt --reps 3 --spec tree --spec "ahead1=libdvbs2_fec_hip_ahead1.so" --spec "ahead2=libdvbs2_fec_hip_ahead2.so" \
  S2_TABLE_C1:25:16384 S2X_TABLE_C1:25:16384 S2X_TABLE_C8:25:8192 S2X_TABLE_C9:25:8192 S2_TABLE_B1:50:4096 S2_TABLE_C2:25:16384 S2_TABLE_C3:25:16384 S2_TABLE_C4:25:16384 S2X_TABLE_C3:25:16384 S2X_TABLE_C10:25:8192
for L2 in "" ahead1; do
  DVBS2_LIB=$PWD/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip${L2:+_$L2}.so python tools/exp_awgn2.py 16384 0.5 7 S2_TABLE_C1 25 2>&1 | tail -1 | sed "s/^/[${L2:-tree}] /"
done > $O/awgn.txt 2>&1; cat $O/awgn.txt
