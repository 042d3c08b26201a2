#!/bin/bash
# round 5, call 10: config3_awgn per-call times; lane-chain block limit per class (DVBS2_LANE_CHAIN_MAX) and the lane chain at degree 30
O=gpurun_out/r5j; mkdir -p $O
python tools/exp_chain_awgn.py 2>&1 | tee $O/chain_awgn.txt | tail -6
python tools/abx.py --out $O/lcmax.txt --spec tree --spec "lc32=,DVBS2_LANE_CHAIN_MAX=32" --spec "lc64=,DVBS2_LANE_CHAIN_MAX=64" --spec "lc100=,DVBS2_LANE_CHAIN_MAX=100" \
  S2_TABLE_B8:50:4096 S2_TABLE_B9:50:4096 S2_TABLE_B10:50:4096
python tools/abx.py --out $O/lc30.txt --spec tree --spec "lc30_180=libdvbs2_fec_hip_lc30.so" --spec "lc30_48=libdvbs2_fec_hip_lc30.so,DVBS2_LANE_CHAIN_MAX=48" S2_TABLE_B11:50:4096
