#!/bin/bash
# tools/pr_sweep.sh -- GPU box: classic vs parity-in-records sweep kernel on every table eligible for the latter
# (4096 frames, cap 20, noise input). Feeds the selection policy in csrc/ldpc_hip.hip.
for t in S2_TABLE_B1 S2_TABLE_B2 S2_TABLE_B3 S2_TABLE_B4 S2X_TABLE_B1 S2X_TABLE_B2 S2X_TABLE_B3 S2X_TABLE_C1 S2X_TABLE_C2 S2X_TABLE_C3 S2X_TABLE_C8 S2X_TABLE_C9 S2X_TABLE_C10 S2_TABLE_C1 S2_TABLE_C2 S2_TABLE_C3 S2_TABLE_C4; do
  a=$(DVBS2_PR=0 python tools/exp_tables.py $t:20:4096 2>&1 | tail -1 | grep -o "[0-9]* fr/s")
  b=$(DVBS2_PR=1 python tools/exp_tables.py $t:20:4096 2>&1 | tail -1 | grep -o "[0-9]* fr/s")
  echo "$t classic $a | parity-in-records $b"
done
