#!/bin/bash
# round 5, call 8: packed first / last phase also in the hazard layers of the classes 12 / 16 (multi-pair layers; single pairs keep the packed chain node)
O=gpurun_out/r5h; mkdir -p $O
python tools/abx.py --out $O/v2p12.txt --spec tree --spec "v2p12=libdvbs2_fec_hip_v2p12.so" \
  S2_TABLE_B5:50:4096 S2_TABLE_B7:50:4096 S2_TABLE_B6:50:4096 S2X_TABLE_B8:50:4096 S2X_TABLE_B6:50:4096 S2_TABLE_C7:25:16384 S2_TABLE_C8:25:16384 T2_TABLE_A3:50:4096
