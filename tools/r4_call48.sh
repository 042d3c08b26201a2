#!/bin/bash
O=gpurun_out/r4au; mkdir -p $O
L=$PWD/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip_cb180.so
DVBS2_LIB=$L DVBS2_LANE_CHAIN_MAX=180 timeout 1500 python -m pytest tests/test_ldpc_gpu.py -m gpu -x -q -k "every_table_bit_exact and (policy or plain or packed) or near_threshold or never" 2>&1 | tail -1 > $O/test.txt; cat $O/test.txt
for rep in 1 2; do
for v in tree cb180; do
  if [ $v = tree ]; then E="X=1"; else E="DVBS2_LIB=$L DVBS2_LANE_CHAIN_MAX=180"; fi
  echo "== $v" >> $O/res.txt
  env $E python tools/exp_tables.py S2_TABLE_B4:50:4096 S2_TABLE_B2:50:4096 S2_TABLE_B3:50:4096 S2_TABLE_B5:50:4096 S2_TABLE_B7:50:4096 S2_TABLE_B8:50:4096 S2_TABLE_B11:50:4096 S2_TABLE_C3:25:16384 2>&1 | grep fr/s | cut -c1-90 >> $O/res.txt
done; done
cat $O/res.txt
