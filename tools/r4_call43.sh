#!/bin/bash
O=gpurun_out/r4ap; mkdir -p $O
bash tools/ab3.sh $O/ab.log "libdvbs2_fec_hip.so libdvbs2_fec_hip_pf12.so libdvbs2_fec_hip_pf12pab.so libdvbs2_fec_hip_pab16.so" S2_TABLE_B4:50:4096 S2_TABLE_B5:50:4096 S2_TABLE_B6:50:4096 S2X_TABLE_B4:50:4096 S2X_TABLE_B6:50:4096 S2X_TABLE_B14:50:4096 S2X_TABLE_B22:50:4096 T2_TABLE_A3:50:4096 S2_TABLE_C6:25:16384 S2X_TABLE_C5:25:16384 S2_TABLE_B7:50:4096 S2X_TABLE_B9:50:4096 S2X_TABLE_B17:50:4096 S2_TABLE_C7:25:16384 > $O/ab_res.txt 2>&1
cat $O/ab_res.txt
