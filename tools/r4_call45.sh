#!/bin/bash
# the parity-in-records tables on the sweep kernel's builds instead (DVBS2_PR=0), now that those gained 10-20 %
O=gpurun_out/r4ar; mkdir -p $O
python tools/exp_tables.py S2_TABLE_C1:25:16384 S2_TABLE_C2:25:16384 S2_TABLE_C3:25:16384 S2_TABLE_C4:25:16384 S2X_TABLE_C1:25:16384 S2X_TABLE_C2:25:16384 S2X_TABLE_C3:25:16384 S2X_TABLE_C8:25:16384 S2X_TABLE_C9:25:16384 S2X_TABLE_C10:25:16384 2>&1 | grep fr/s | cut -c1-95 > $O/pr.txt
DVBS2_PR=0 python tools/policy_sweep.py S2_TABLE_C1 S2_TABLE_C2 S2_TABLE_C3 S2_TABLE_C4 S2X_TABLE_C1 S2X_TABLE_C2 S2X_TABLE_C3 S2X_TABLE_C8 S2X_TABLE_C9 S2X_TABLE_C10 > $O/nopr.txt 2>&1
cat $O/pr.txt $O/nopr.txt
