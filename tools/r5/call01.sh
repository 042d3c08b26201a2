#!/bin/bash
# round 5, call 1: timing-only bounds (wrong results by construction) -- VERDICT r4 items 1 and 3
O=gpurun_out/r5a; mkdir -p $O
python tools/abx.py --out $O/oneword.txt --spec tree --spec "ow=libdvbs2_fec_hip_oneword.so" \
  S2_TABLE_B4:50:4096 S2X_TABLE_B3:50:4096 S2_TABLE_B7:50:4096 S2X_TABLE_B8:50:4096 S2_TABLE_B5:50:4096 S2X_TABLE_B21:50:4096 S2_TABLE_B11:50:4096
python tools/abx.py --out $O/nohaz.txt --spec tree --spec "nh=,DVBS2_EXP_NOHAZ=1,DVBS2_SOFT_BARRIER=0" \
  --spec "nhv2=,DVBS2_EXP_NOHAZ=1,DVBS2_V2=1,DVBS2_SOFT_BARRIER=0" --spec "nhsoft=,DVBS2_EXP_NOHAZ=1" --spec "nhv2soft=,DVBS2_EXP_NOHAZ=1,DVBS2_V2=1" \
  S2_TABLE_B11:50:4096 S2_TABLE_B10:50:4096 S2_TABLE_B9:50:4096 S2_TABLE_B8:50:4096 S2_TABLE_B7:50:4096
