#!/bin/bash
O=gpurun_out/r4ai; mkdir -p $O
DVBS2_LIB=$PWD/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip_prpab.so timeout 1500 python -m pytest tests/test_ldpc_gpu.py -m gpu -x -q -k "every_table_bit_exact and policy and (C1 or C2 or C3 or C4 or C8 or C9 or C10) or parity_in_records" 2>&1 | tail -1 > $O/test.txt; cat $O/test.txt
bash tools/ab3.sh $O/ab.log "libdvbs2_fec_hip.so libdvbs2_fec_hip_prpab.so" S2_TABLE_C1:25:16384 S2_TABLE_C2:25:16384 S2_TABLE_C3:25:16384 S2_TABLE_C4:25:16384 S2X_TABLE_C1:25:16384 S2X_TABLE_C3:25:16384 S2X_TABLE_C8:25:16384 S2X_TABLE_C10:25:16384 > $O/ab_res.txt 2>&1
cat $O/ab_res.txt
