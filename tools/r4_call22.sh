#!/bin/bash
# decoupled frames with a start offset (DVBS2_EXP_STAGGER build, software frame barriers, two-level lane chain kept)
O=gpurun_out/r4v; mkdir -p $O
L=$PWD/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip_stag.so
DVBS2_LIB=$L DVBS2_SOFT_BARRIER=1 DVBS2_STAGGER=37 timeout 900 python -m pytest tests/test_ldpc_gpu.py -m gpu -x -q -k "every_table_bit_exact and (B11 or B9 or B21) and policy" 2>&1 | tail -2 > $O/test.txt
cat $O/test.txt
for rep in 1 2; do
  echo "== tree default" >> $O/stag.txt
  python tools/exp_tables.py S2_TABLE_B11:50:4096 S2_TABLE_B9:50:4096 2>&1 | grep fr/s >> $O/stag.txt
  for st in 0 10 20 30 37 45 60; do
    echo "== soft+tlc stagger $st" >> $O/stag.txt
    DVBS2_LIB=$L DVBS2_SOFT_BARRIER=1 DVBS2_STAGGER=$st python tools/exp_tables.py S2_TABLE_B11:50:4096 S2_TABLE_B9:50:4096 2>&1 | grep fr/s >> $O/stag.txt
  done
done
cat $O/stag.txt
