#!/bin/bash
O=gpurun_out/r4m; mkdir -p $O
for lib in libdvbs2_fec_hip_fw0.so libdvbs2_fec_hip.so; do
  DVBS2_LIB=$PWD/gr-dvbs2rx_amd/lib/$lib DVBS2_TIMING=1 DVBS2_TIMING_LAYERS=1 python tools/exp_tables.py S2_TABLE_B4:10:512 > $O/timing_B4_$lib.log 2>&1
  echo "== $lib"; grep -a "hazard phases\|block  [1-9]\|block [1-9][0-9] \|block 1[0-9][0-9]\|timing," $O/timing_B4_$lib.log | tail -14
done
