#!/bin/bash
# GPU box, round 6 call 20: B4 / S2X 9/20 through the packed parity-in-records kernel against the classic packed one-frame build: never-converging input and the operating point
O=gpurun_out/r6w; mkdir -p $O
timeout 900 python tools/abx.py --out $O/ab.txt --reps 3 --spec tree --spec "prv2_normal=,DVBS2_PR=1,DVBS2_PR_V2=1" S2_TABLE_B4:50:4096 S2X_TABLE_B3:50:4096 S2X_TABLE_B11:50:4096 S2_TABLE_B3:50:4096 S2X_TABLE_B2:50:4096
for r in 1 2; do
  python tools/exp_awgn2.py 4096 2.0 7 S2_TABLE_B4 50 2>&1 | tail -1 | sed "s/^/[tree] /"
  DVBS2_PR=1 DVBS2_PR_V2=1 python tools/exp_awgn2.py 4096 2.0 7 S2_TABLE_B4 50 2>&1 | tail -1 | sed "s/^/[prv2] /"
done > $O/awgn.txt 2>&1; cat $O/awgn.txt
