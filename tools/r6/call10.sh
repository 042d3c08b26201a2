#!/bin/bash
# GPU box, round 6 call 10: the parity-in-records kernel on NORMAL frames of check degree <= 7 (four frames per CU) against the policy's classic builds
O=gpurun_out/r6j; mkdir -p $O
timeout 1500 python tools/abx.py --out $O/ab_pr_normal.txt --reps 2 --spec tree --spec "pr=,DVBS2_PR=1" --spec "pr_bytes=,DVBS2_PR=1,DVBS2_PR_W1=0" \
  S2_TABLE_B1:50:4096 S2_TABLE_B2:50:4096 S2_TABLE_B3:50:4096 S2_TABLE_B4:50:4096 S2X_TABLE_B1:50:4096 S2X_TABLE_B2:50:4096 S2X_TABLE_B3:50:4096
