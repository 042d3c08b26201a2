#!/bin/bash
O=gpurun_out/r4j; mkdir -p $O
for t in S2_TABLE_B11 S2_TABLE_B7 S2_TABLE_B4 S2X_TABLE_B21; do
  DVBS2_LIB=$PWD/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip_timing.so DVBS2_TIMING=1 DVBS2_TIMING_LAYERS=1 DVBS2_TIMING_WAVES=1 python tools/exp_tables.py $t:10:512 > $O/timing_$t.log 2>&1
done
tail -40 $O/timing_S2_TABLE_B11.log
