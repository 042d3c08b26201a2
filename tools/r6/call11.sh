#!/bin/bash
# GPU box, round 6 call 11: parity-in-records kernel with per-wave one-add addresses -- bit-exactness and A/B
O=gpurun_out/r6k; mkdir -p $O
timeout 1200 python -m pytest tests/test_ldpc_gpu.py -x -q -n 4 -k "policy or pr-byte or C1 or C4 or group or near or saturation or counters or config1 or baseline" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
DVBS2_PR=1 timeout 600 python -m pytest tests/test_ldpc_gpu.py -x -q -n 4 -k "test_every_table_bit_exact and policy and (S2_TABLE_B1 or S2_TABLE_B2 or S2_TABLE_B3 or S2_TABLE_B4 or S2X_TABLE_B1 or S2X_TABLE_B2 or S2X_TABLE_B3)" > $O/pytest_prn.log 2>&1; echo "pytest pr-normal rc $?"; tail -2 $O/pytest_prn.log
timeout 2400 python tools/abx.py --out $O/ab.txt --reps 3 --spec "base=libdvbs2_fec_hip_base.so" --spec tree --spec "tree_pw0=,DVBS2_PR_PW=0" \
  S2_TABLE_C1:25:16384 S2_TABLE_C2:25:16384 S2_TABLE_C3:25:16384 S2_TABLE_C4:25:16384 S2X_TABLE_C1:25:16384 S2X_TABLE_C2:25:16384 S2X_TABLE_C3:25:16384 S2X_TABLE_C8:25:8192 S2X_TABLE_C9:25:8192 S2X_TABLE_C10:25:8192
timeout 1200 python tools/abx.py --out $O/ab_normal.txt --reps 3 --spec tree --spec "pr=,DVBS2_PR=1" --spec "pr_pw0=,DVBS2_PR=1,DVBS2_PR_PW=0" \
  S2_TABLE_B1:50:4096 S2_TABLE_B2:50:4096 S2_TABLE_B3:50:4096 S2_TABLE_B4:50:4096 S2X_TABLE_B1:50:4096 S2X_TABLE_B3:50:4096
