#!/bin/bash
O=gpurun_out/r4r; mkdir -p $O
bash tools/ab3.sh $O/ab3.log "libdvbs2_fec_hip.so libdvbs2_fec_hip_fw16.so" S2_TABLE_B7:50:4096 S2X_TABLE_B8:50:4096 S2X_TABLE_B16:50:4096 S2X_TABLE_B23:50:4096 S2_TABLE_C7:25:16384 S2_TABLE_C8:25:16384 S2X_TABLE_C7:25:16384 S2_TABLE_B6:50:4096 S2X_TABLE_B4:50:4096 S2X_TABLE_B12:50:4096 S2_TABLE_C5:25:16384 > $O/ab3_res.log 2>&1
cat $O/ab3_res.log
for t in S2_TABLE_B5 T2_TABLE_A3 S2X_TABLE_C5 T2_TABLE_B3; do for lib in libdvbs2_fec_hip.so libdvbs2_fec_hip_fw16.so; do echo -n "$t plain+solo $lib: "; DVBS2_V2=0 DVBS2_CHAIN_V2=0 DVBS2_SOLO=1 DVBS2_LIB=$PWD/gr-dvbs2rx_amd/lib/$lib python tools/exp_tables.py $t:$([ ${t:0:5} = S2X_T -o ${t:0:4} = T2_T ] && echo 25:16384 || echo 50:4096) 2>/dev/null | awk '{print $8, $9}'; done; done
