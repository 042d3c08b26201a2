#!/bin/bash
# GPU box, round 6 call 3: route (ii) bound -- one-frame workgroups (128 VGPRs, whatever the compiler spills) for the degree classes 20-32 --
# and the two-barrier lane chain in the packed hazard nodes of degree > 20
O=gpurun_out/r6c; mkdir -p $O
timeout 1500 python tools/abx.py --out $O/ab_solo.txt --reps 2 --spec tree --spec "solo_pk=libdvbs2_fec_hip_solo32.so,DVBS2_SOLO=1,DVBS2_V2=1" --spec "solo_plain=libdvbs2_fec_hip_solo32.so,DVBS2_SOLO=1,DVBS2_V2=0" \
  S2_TABLE_B11:50:4096 S2_TABLE_B10:50:4096 S2_TABLE_B9:50:4096 S2_TABLE_B8:50:4096 S2X_TABLE_B21:50:4096 S2X_TABLE_B10:50:4096
timeout 1200 python tools/abx.py --out $O/ab_tb.txt --reps 3 --spec tree --spec "tb=libdvbs2_fec_hip_tb.so" \
  S2_TABLE_B11:50:4096 S2_TABLE_B10:50:4096 S2_TABLE_B9:50:4096 S2_TABLE_B8:50:4096 S2X_TABLE_B21:50:4096 S2_TABLE_C10:25:8192
DVBS2_LIB=$PWD/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip_tb.so timeout 600 python -m pytest tests/test_ldpc_gpu.py -x -q -k "test_every_table_bit_exact and (packed-pair or policy) and (S2_TABLE_B8 or S2_TABLE_B9 or S2_TABLE_B10 or S2_TABLE_B11 or S2_TABLE_C9 or S2_TABLE_C10)" > $O/pytest_tb.log 2>&1; echo "pytest tb rc $?"; tail -2 $O/pytest_tb.log
