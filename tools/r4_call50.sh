#!/bin/bash
O=gpurun_out/r4aw; mkdir -p $O
bash tools/ab3.sh $O/ab.log "libdvbs2_fec_hip.so libdvbs2_fec_hip_p023.so libdvbs2_fec_hip_p003.so libdvbs2_fec_hip_p123.so libdvbs2_fec_hip_p012.so libdvbs2_fec_hip_p013b.so" S2_TABLE_B4:50:4096 S2_TABLE_B7:50:4096 S2X_TABLE_B21:50:4096 S2_TABLE_B5:50:4096 S2X_TABLE_B3:50:4096 S2_TABLE_C7:25:16384 > $O/ab_res.txt 2>&1
cat $O/ab_res.txt
