#!/bin/bash
# GPU box, round 6 call 2: cycle stamps per layer / hazard phase, block steps against wide steps
O=gpurun_out/r6b; mkdir -p $O
L=$PWD/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip_timing.so
for T in S2_TABLE_B11 S2_TABLE_B10 S2_TABLE_B8; do
for W in 0 1; do
  DVBS2_LIB=$L DVBS2_TIMING=1 DVBS2_TIMING_LAYERS=1 DVBS2_V2=1 DVBS2_WIDE=$W WARM_S=0.05 timeout 300 python tools/exp_tables.py $T:10:512 > $O/stamps_${T}_w$W.txt 2>&1
done; done
tail -30 $O/stamps_S2_TABLE_B11_w1.txt
