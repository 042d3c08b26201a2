#!/bin/bash
O=gpurun_out/r4ax; mkdir -p $O
DVBS2_LIB=$PWD/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip_lean.so timeout 1500 python -m pytest tests/test_ldpc_gpu.py -m gpu -x -q -k "every_table_bit_exact or near_threshold or never or full_batch or ragged or group_of_one" 2>&1 | tail -1 > $O/test.txt; cat $O/test.txt
bash tools/ab3.sh $O/ab.log "libdvbs2_fec_hip.so libdvbs2_fec_hip_lean.so" S2_TABLE_B4:50:4096 S2_TABLE_B3:50:4096 S2X_TABLE_B3:50:4096 S2_TABLE_B1:50:4096 S2_TABLE_B2:50:4096 S2_TABLE_B5:50:4096 S2_TABLE_B7:50:4096 S2_TABLE_B11:50:4096 S2X_TABLE_B21:50:4096 S2_TABLE_C7:25:16384 S2_TABLE_C5:25:16384 S2X_TABLE_B8:50:4096 > $O/ab_res.txt 2>&1
cat $O/ab_res.txt
