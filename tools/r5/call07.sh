#!/bin/bash
# round 5, call 7: "pure" packed builds of the classes 24-32 (plain nodes compiled for layer 0 only) against the tree
O=gpurun_out/r5g; mkdir -p $O
python tools/abx.py --out $O/pure.txt --spec tree --spec "v2p=,DVBS2_V2=1" --spec "pure=libdvbs2_fec_hip_pure.so,DVBS2_V2=1" \
  S2_TABLE_B11:50:4096 S2_TABLE_B10:50:4096 S2_TABLE_B9:50:4096
