#!/bin/bash
# tools/quick_r2.sh <tag> [tables...] -- GPU box: LDPC parity tests + frames/s of the BASELINE tables (noise input, full cap)
TAG=${1:-q}; shift
mkdir -p gpurun_out/$TAG
python -m pytest tests/test_ldpc_gpu.py -x -q -m gpu 2>&1 | tail -3 | tee gpurun_out/$TAG/pytest.log
TABLES=${@:-S2_TABLE_B4:50:4096 S2_TABLE_B7:50:4096 S2_TABLE_B11:50:4096 S2X_TABLE_B21:50:4096 S2_TABLE_C1:25:16384}
python tools/exp_tables.py $TABLES 2>&1 | grep -v amdgpu.ids | tee gpurun_out/$TAG/tables.log
