#!/bin/bash
O=gpurun_out/r4al; mkdir -p $O
timeout 2400 python -m pytest tests/test_ldpc_gpu.py tests/test_bch_demap_gpu.py -m gpu -x -q 2>&1 | tail -1 > $O/test.txt; cat $O/test.txt
timeout 900 python tools/soft_sweep.py > $O/soft_sweep.txt 2>&1; cat $O/soft_sweep.txt
for e in "X=1" "DVBS2_SOFT_BARRIER=0" "DVBS2_SOFT_BARRIER=1"; do echo "== $e" >> $O/soft_free.txt; env $e python tools/exp_tables.py S2X_TABLE_B21:50:4096 S2X_TABLE_B10:50:4096 S2X_TABLE_B19:50:4096 S2X_TABLE_B20:50:4096 S2X_TABLE_B24:50:4096 2>&1 | grep fr/s | cut -c1-90 >> $O/soft_free.txt; done; cat $O/soft_free.txt
