#!/bin/bash
# one lease of the policy sweep (tables that run the sweep kernel; the parity-in-records tables have no choice)
O=gpurun_out/r4pol; mkdir -p $O
T="S2_TABLE_B1 S2_TABLE_B2 S2_TABLE_B3 S2_TABLE_B4 S2_TABLE_B5 S2_TABLE_B6 S2_TABLE_B7 S2_TABLE_B8 S2_TABLE_B9 S2_TABLE_B10 S2_TABLE_B11 S2_TABLE_C5 S2_TABLE_C6 S2_TABLE_C7 S2_TABLE_C8 S2_TABLE_C9 S2_TABLE_C10"
for i in $(seq 1 24); do T="$T S2X_TABLE_B$i"; done
T="$T S2X_TABLE_C4 S2X_TABLE_C5 S2X_TABLE_C6 S2X_TABLE_C7 T2_TABLE_A3 T2_TABLE_B3"
python tools/policy_sweep.py $T > $O/sweep_$1.log 2>&1
tail -3 $O/sweep_$1.log
