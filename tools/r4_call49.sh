#!/bin/bash
O=gpurun_out/r4av; mkdir -p $O
bash tools/ab3.sh $O/ab.log "libdvbs2_fec_hip.so libdvbs2_fec_hip_noprio.so" S2_TABLE_B4:50:4096 S2_TABLE_B7:50:4096 S2_TABLE_B11:50:4096 S2X_TABLE_B21:50:4096 S2_TABLE_B1:50:4096 S2_TABLE_B2:50:4096 S2_TABLE_B5:50:4096 S2_TABLE_C1:25:16384 S2_TABLE_C7:25:16384 > $O/ab_res.txt 2>&1
cat $O/ab_res.txt
