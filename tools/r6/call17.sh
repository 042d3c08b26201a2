#!/bin/bash
# GPU box, round 6 call 17: full syndrome test of the parity-in-records kernel with all parity rows of a short frame requested at once (-DDVBS2_PR_SYN_CHUNK=36)
O=gpurun_out/r6t; mkdir -p $O
DVBS2_LIB=$PWD/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip_syn36.so timeout 900 python -m pytest tests/test_ldpc_gpu.py -q -n 4 -k "pr-byte or C1 or C2 or C4 or C8 or group or near or saturation or counters" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
timeout 900 python tools/abx.py --out $O/ab.txt --reps 3 --spec tree --spec "syn36=libdvbs2_fec_hip_syn36.so" S2_TABLE_C1:25:16384 S2_TABLE_C2:25:16384 S2X_TABLE_C9:25:8192 S2_TABLE_B1:50:4096
for r in 1 2; do for L2 in "" syn36; do
  DVBS2_LIB=$PWD/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip${L2:+_$L2}.so python tools/exp_awgn2.py 16384 0.5 7 S2_TABLE_C1 25 2>&1 | tail -1 | sed "s/^/[${L2:-tree}] /"
  DVBS2_LIB=$PWD/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip${L2:+_$L2}.so python tools/exp_awgn2.py 16384 1.5 7 S2_TABLE_C2 25 2>&1 | tail -1 | sed "s/^/[${L2:-tree}] /"
done; done > $O/awgn.txt 2>&1; cat $O/awgn.txt
