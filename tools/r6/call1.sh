#!/bin/bash
# GPU box, round 6 call 1: wide ordered steps -- bit-exactness on the packed builds of the degree classes >= 20, then A/B against the block scheme
O=gpurun_out/r6a; mkdir -p $O
timeout 900 python -m pytest tests/test_ldpc_gpu.py -x -q -k "test_every_table_bit_exact and (packed-pair or policy or soft-barrier) and (S2_TABLE_B8 or S2_TABLE_B9 or S2_TABLE_B10 or S2_TABLE_B11 or S2_TABLE_C9 or S2_TABLE_C10 or S2_TABLE_C5-)" > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
DVBS2_WIDE=2 timeout 600 python -m pytest tests/test_ldpc_gpu.py -x -q -k "test_every_table_bit_exact and (packed-pair) and (S2_TABLE_B8 or S2_TABLE_B9 or S2_TABLE_B10 or S2_TABLE_B11 or S2_TABLE_C9 or S2_TABLE_C10)" > $O/pytest_w2.log 2>&1; echo "pytest w2 rc $?"; tail -3 $O/pytest_w2.log
timeout 1500 python tools/abx.py --out $O/ab.txt --reps 2 --spec "block=,DVBS2_WIDE=0" --spec "wide=,DVBS2_WIDE=1" --spec "wide_tlc=,DVBS2_WIDE=2" \
  S2_TABLE_B11:50:4096 S2_TABLE_B10:50:4096 S2_TABLE_B9:50:4096 S2_TABLE_B8:50:4096 S2X_TABLE_B21:50:4096
