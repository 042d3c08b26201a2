#!/bin/bash
O=gpurun_out/r4o; mkdir -p $O
bash tools/ab3.sh $O/ab3.log "libdvbs2_fec_hip_r3.so libdvbs2_fec_hip_as3fw0.so libdvbs2_fec_hip.so" S2_TABLE_B5:50:4096 S2_TABLE_B6:50:4096 S2_TABLE_C5:25:16384 S2_TABLE_C6:25:16384 T2_TABLE_A3:50:4096 S2_TABLE_B4:50:4096 S2_TABLE_B2:50:4096 S2_TABLE_B7:50:4096 S2X_TABLE_B4:50:4096 S2X_TABLE_B12:50:4096 S2X_TABLE_C5:25:16384 > $O/ab3_res.log 2>&1
cat $O/ab3_res.log
