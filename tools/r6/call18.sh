#!/bin/bash
# GPU box, round 6 call 18: packed nodes in the parity-in-records kernel (DVBS2_PR_V2=1; normal frames with DVBS2_PR=1) -- bit-exactness and A/B
O=gpurun_out/r6u; mkdir -p $O
timeout 1200 python -m pytest tests/test_ldpc_gpu.py -q -n 4 -k "test_every_table_bit_exact and pr-packed" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
DVBS2_PR_V2=1 timeout 900 python -m pytest tests/test_ldpc_gpu.py -q -n 4 -k "(C2 or C3 or C4 or group or near or saturation or counters) and not every_table" > $O/pytest2.log 2>&1; echo "pytest2 rc $?"; tail -2 $O/pytest2.log
timeout 1500 python tools/abx.py --out $O/ab.txt --reps 3 --spec tree --spec "prv2=,DVBS2_PR_V2=1" --spec "pr_normal=,DVBS2_PR=1" --spec "prv2_normal=,DVBS2_PR=1,DVBS2_PR_V2=1" \
  S2_TABLE_B4:50:4096 S2_TABLE_B3:50:4096 S2_TABLE_B2:50:4096 S2X_TABLE_B3:50:4096 S2_TABLE_C2:25:16384 S2_TABLE_C3:25:16384 S2_TABLE_C4:25:16384 S2X_TABLE_C3:25:16384 S2X_TABLE_C10:25:8192
