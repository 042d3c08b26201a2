#!/bin/bash
O=gpurun_out/r4i; mkdir -p $O
bash tools/ab_tables.sh $O/ab.log libdvbs2_fec_hip.so libdvbs2_fec_hip_nowrap.so S2X_TABLE_B21:50:4096 S2_TABLE_B11:50:4096 S2_TABLE_B4:50:4096 S2X_TABLE_B3:50:4096 S2_TABLE_B9:50:4096 S2X_TABLE_B10:50:4096 > $O/ab_res.log 2>&1
cat $O/ab_res.log
