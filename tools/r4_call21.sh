#!/bin/bash
# placement of 6-wave workgroups with the occupancy capped at 3 by the VGPR allocation; decoupled frames (software frame barriers) on 9/10 normal
O=gpurun_out/r4u; mkdir -p $O
tools/bin/placement 6 75000 0 > $O/placement.txt 2>&1
tools/bin/placement 6 75000 1 >> $O/placement.txt 2>&1
cat $O/placement.txt
for rep in 1 2; do
for e in "X=1" "DVBS2_TWO_LEVEL=0" "DVBS2_SOFT_BARRIER=1" "DVBS2_SOFT_BARRIER=1 DVBS2_TWO_LEVEL=0"; do
  echo "== $e" >> $O/soft.txt
  env $e python tools/exp_tables.py S2_TABLE_B11:50:4096 S2_TABLE_B9:50:4096 S2_TABLE_B10:50:4096 2>&1 | grep fr/s >> $O/soft.txt
done; done
cat $O/soft.txt
