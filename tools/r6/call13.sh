#!/bin/bash
# GPU box, round 6 call 13: degree-3 / 4 node of the one-dword-record kernel with the minimum over the OTHER links taken directly
# (-DDVBS2_PR6_DIRECT=1, lib/libdvbs2_fec_hip_direct.so) -- bit-exactness and A/B against the tree
O=gpurun_out/r6m; mkdir -p $O
L=$PWD/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip_direct.so
DVBS2_LIB=$L timeout 1200 python -m pytest tests/test_ldpc_gpu.py -x -q -n 4 -k "policy or pr-byte or C1 or C4 or C8 or C9 or group or near or saturation or counters or config1 or baseline or (test_every_table_bit_exact and (S2_TABLE_B1- or S2X_TABLE_B1-))" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
timeout 1500 python tools/abx.py --out $O/ab.txt --reps 3 --spec tree --spec "direct=libdvbs2_fec_hip_direct.so" \
  S2_TABLE_C1:25:16384 S2X_TABLE_C1:25:16384 S2X_TABLE_C8:25:8192 S2X_TABLE_C9:25:8192 S2_TABLE_B1:50:4096 S2X_TABLE_B1:50:4096 S2_TABLE_C2:25:16384 S2_TABLE_C4:25:16384
for L2 in "" direct; do
  DVBS2_LIB=$PWD/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip${L2:+_$L2}.so python tools/exp_awgn2.py 16384 0.5 7 S2_TABLE_C1 25 2>&1 | tail -1 | sed "s/^/[${L2:-tree}] /"
done > $O/awgn.txt 2>&1; cat $O/awgn.txt
