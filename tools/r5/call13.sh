#!/bin/bash
# round 5, call 13: two's complement LDS in the class-8 packed builds only (tree) vs none (notc): the class's packed tables; then the whole suite
O=gpurun_out/r5m; mkdir -p $O
python tools/abx.py --out $O/tc8.txt --spec "notc=libdvbs2_fec_hip_notc.so" --spec tree \
  S2_TABLE_B4:50:4096 S2_TABLE_B3:50:4096 S2X_TABLE_B3:50:4096 S2X_TABLE_B11:50:4096 S2_TABLE_B7:50:4096 S2_TABLE_B8:50:4096
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_gpu.log
