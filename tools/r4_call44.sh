#!/bin/bash
O=gpurun_out/r4aq; mkdir -p $O
DVBS2_LIB=$PWD/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip_ldsbar.so timeout 900 python -m pytest tests/test_ldpc_gpu.py -m gpu -x -q -k "every_table_bit_exact and policy and (B9 or B10 or B11 or B8 or C9 or C10)" 2>&1 | tail -1 > $O/test.txt; cat $O/test.txt
bash tools/ab3.sh $O/ab.log "libdvbs2_fec_hip.so libdvbs2_fec_hip_ldsbar.so" S2_TABLE_B11:50:4096 S2_TABLE_B10:50:4096 S2_TABLE_B9:50:4096 S2_TABLE_B8:50:4096 S2_TABLE_C9:25:16384 S2_TABLE_C10:25:16384 S2_TABLE_B7:50:4096 S2_TABLE_B5:50:4096 > $O/ab_res.txt 2>&1
cat $O/ab_res.txt
