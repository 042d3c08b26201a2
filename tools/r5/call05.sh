#!/bin/bash
# round 5, call 5: hazard layers with the packed first / last phase (V2P): bit-exactness of the packed builds on every table, then A/B
O=gpurun_out/r5e; mkdir -p $O
timeout 900 python -m pytest tests/test_ldpc_gpu.py -x -q -k "test_every_table_bit_exact and (packed or soft)" > $O/pytest_v2p.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest_v2p.log
python tools/abx.py --out $O/v2p.txt --spec tree --spec "v2p=,DVBS2_V2=1" --spec "v2plainhz=,DVBS2_V2=1,DVBS2_V2P=0" --spec "plain=,DVBS2_V2=0" \
  S2_TABLE_B11:50:4096 S2_TABLE_B10:50:4096 S2_TABLE_B9:50:4096 S2_TABLE_B8:50:4096 S2X_TABLE_B21:50:4096
