#!/bin/bash
# GPU box, round 6 call 15: record accesses of the parity-in-records loop as scalar base + loop-invariant per-lane byte offset (-DDVBS2_PR_SADDR=1) -- bit-exactness and A/B
O=gpurun_out/r6p; mkdir -p $O
SEL="pr-byte or C1 or C2 or C4 or C8 or group or near or saturation or counters"
DVBS2_LIB=$PWD/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip_saddr.so timeout 900 python -m pytest tests/test_ldpc_gpu.py -q -n 4 -k "$SEL" > $O/pytest_saddr.log 2>&1; echo "pytest saddr rc $?"; tail -2 $O/pytest_saddr.log
timeout 1200 python tools/abx.py --out $O/ab.txt --reps 3 --spec tree --spec "saddr=libdvbs2_fec_hip_saddr.so" \
  S2_TABLE_C1:25:16384 S2X_TABLE_C1:25:16384 S2X_TABLE_C9:25:8192 S2_TABLE_B1:50:4096 S2_TABLE_C2:25:16384 S2_TABLE_C3:25:16384 S2_TABLE_C4:25:16384
