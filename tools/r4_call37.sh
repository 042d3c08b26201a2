#!/bin/bash
O=gpurun_out/r4aj; mkdir -p $O
L=$PWD/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip_ptc.so
DVBS2_LIB=$L timeout 1500 python -m pytest tests/test_ldpc_gpu.py -m gpu -x -q -k "every_table_bit_exact and policy or near_threshold or never or full_batch or group_stop or digests" 2>&1 | tail -1 > $O/test.txt; cat $O/test.txt
for rep in 1 2 3; do
  python tools/exp_awgn2.py 4096 2.0 6 2>/dev/null | tail -1 >> $O/awgn.txt
  DVBS2_LIB=$L python tools/exp_awgn2.py 4096 2.0 6 2>/dev/null | tail -1 | sed 's/^/PTC /' >> $O/awgn.txt
done
cat $O/awgn.txt
bash tools/ab3.sh $O/ab.log "libdvbs2_fec_hip.so libdvbs2_fec_hip_ptc.so" S2_TABLE_B4:50:4096 S2_TABLE_B7:50:4096 S2_TABLE_B11:50:4096 S2X_TABLE_B21:50:4096 S2_TABLE_B1:50:4096 S2_TABLE_C7:25:16384 > $O/ab_res.txt 2>&1
cat $O/ab_res.txt
