#!/bin/bash
O=gpurun_out/r4y; mkdir -p $O
for v in ep wa epwa; do
  DVBS2_LIB=$PWD/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip_$v.so timeout 1200 python -m pytest tests/test_ldpc_gpu.py -m gpu -x -q -k "every_table_bit_exact and (policy or plain) or near_threshold or never_conv" 2>&1 | tail -1 >> $O/test.txt
done
cat $O/test.txt
bash tools/ab3.sh $O/ab.log "libdvbs2_fec_hip.so libdvbs2_fec_hip_ep.so libdvbs2_fec_hip_wa.so libdvbs2_fec_hip_epwa.so" S2_TABLE_B4:50:4096 S2_TABLE_B2:50:4096 S2_TABLE_B3:50:4096 S2_TABLE_B1:50:4096 S2_TABLE_B5:50:4096 S2_TABLE_B6:50:4096 T2_TABLE_A3:50:4096 S2_TABLE_C3:25:16384 S2_TABLE_C7:25:16384 S2_TABLE_B7:50:4096 > $O/ab_res.txt 2>&1
cat $O/ab_res.txt
