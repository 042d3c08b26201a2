#!/bin/bash
O=gpurun_out/r4ag; mkdir -p $O
DVBS2_LIB=$PWD/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip_pf0.so timeout 1500 python -m pytest tests/test_ldpc_gpu.py -m gpu -x -q -k "every_table_bit_exact and policy or near_threshold or never" 2>&1 | tail -1 > $O/test.txt; cat $O/test.txt
bash tools/ab3.sh $O/ab.log "libdvbs2_fec_hip.so libdvbs2_fec_hip_pf0.so" S2_TABLE_B7:50:4096 S2_TABLE_B8:50:4096 S2_TABLE_B9:50:4096 S2_TABLE_B10:50:4096 S2_TABLE_B11:50:4096 S2X_TABLE_B21:50:4096 S2X_TABLE_B8:50:4096 S2X_TABLE_B16:50:4096 S2X_TABLE_B10:50:4096 S2X_TABLE_B19:50:4096 S2X_TABLE_B20:50:4096 S2X_TABLE_B24:50:4096 S2_TABLE_C7:25:16384 S2_TABLE_C8:25:16384 S2_TABLE_C9:25:16384 S2_TABLE_C10:25:16384 S2X_TABLE_C7:25:16384 > $O/ab_res.txt 2>&1
cat $O/ab_res.txt
