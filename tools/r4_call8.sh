#!/bin/bash
O=gpurun_out/r4h; mkdir -p $O
bash tools/ab_tables.sh $O/ab.log libdvbs2_fec_hip_stag0.so libdvbs2_fec_hip_stag2.so S2_TABLE_B11:50:4096 S2_TABLE_B5:50:4096 S2_TABLE_B10:50:4096 S2_TABLE_B6:50:4096 > $O/ab_res.log 2>&1
cat $O/ab_res.log
