#!/bin/bash
# round 5, call 12: LLR bytes in LDS as two's complement in the packed builds (tree) against offset binary (notc): bit-exactness, then A/B
O=gpurun_out/r5l; mkdir -p $O
timeout 900 python -m pytest tests/test_ldpc_gpu.py tests/test_bch_demap_gpu.py -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -4 $O/pytest.log
python tools/abx.py --out $O/tc.txt --spec "notc=libdvbs2_fec_hip_notc.so" --spec tree \
  S2_TABLE_B4:50:4096 S2X_TABLE_B3:50:4096 S2_TABLE_B7:50:4096 S2_TABLE_B5:50:4096 S2X_TABLE_B8:50:4096 S2X_TABLE_B6:50:4096 S2_TABLE_B11:50:4096 S2_TABLE_B9:50:4096 S2_TABLE_B8:50:4096 S2X_TABLE_B21:50:4096 S2_TABLE_C7:25:16384 S2X_TABLE_C6:25:16384 T2_TABLE_B3:25:16384 S2X_TABLE_B20:50:4096
