#!/bin/bash
O=gpurun_out/r4s; mkdir -p $O
DVBS2_LIB=$PWD/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip_ldsbar.so python -m pytest tests/test_ldpc_gpu.py -m gpu -x -q -k "every_table_bit_exact and policy or near_threshold or never" 2>&1 | tail -2
bash tools/ab3.sh $O/ab3.log "libdvbs2_fec_hip.so libdvbs2_fec_hip_ldsbar.so" S2_TABLE_B4:50:4096 S2_TABLE_B7:50:4096 S2_TABLE_B11:50:4096 S2X_TABLE_B21:50:4096 S2_TABLE_C1:25:16384 S2_TABLE_B2:50:4096 S2_TABLE_B5:50:4096 S2_TABLE_B9:50:4096 S2X_TABLE_B8:50:4096 S2X_TABLE_B4:50:4096 S2_TABLE_C7:25:16384 > $O/ab3_res.log 2>&1
cat $O/ab3_res.log
