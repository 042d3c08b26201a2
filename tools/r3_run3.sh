#!/bin/bash
O=gpurun_out/r3c; mkdir -p $O; L=$PWD/gr-dvbs2rx_amd/lib
timeout 300 python bench.py --only config2_awgn --no-cpu-baseline > $O/bench_awgn.log 2>&1
timeout 900 python -m pytest tests/test_ldpc_gpu.py -x -q -m gpu -k "near_threshold or group_of_one or async or multi_chunk or parity_in_records or ragged or enqueue_finish or config1 or not_the_table_k" 2>&1 | tail -3 > $O/pytest_gs.log
T="S2_TABLE_B11:50:4096"
for env in "" "DVBS2_HZ2=1" "DVBS2_LANE_CHAIN_MAX=0" "DVBS2_HZ2=1 DVBS2_LANE_CHAIN_MAX=0"; do
  echo "== lr $env" >> $O/b11.log; env $env DVBS2_LIB=$L/libdvbs2_fec_hip_lr.so timeout 300 python tools/exp_tables.py $T S2_TABLE_B9:50:4096 S2_TABLE_C10:25:16384 2>&1 | grep -v amdgpu.ids >> $O/b11.log
done
echo "== default lib" >> $O/b11.log; timeout 300 python tools/exp_tables.py $T S2_TABLE_B9:50:4096 S2_TABLE_C10:25:16384 2>&1 | grep -v amdgpu.ids >> $O/b11.log
tm() { echo "== $*" >> $O/timing.log; env "$@" DVBS2_TIMING=1 DVBS2_TIMING_LAYERS=1 timeout 300 python tools/exp_tables.py $TT:10:512 2>&1 | grep -v amdgpu.ids | tail -100 >> $O/timing.log; }
TT=S2_TABLE_B11 tm DVBS2_LIB=$L/libdvbs2_fec_hip_lrt.so DVBS2_V2=0
TT=S2_TABLE_B11 tm DVBS2_LIB=$L/libdvbs2_fec_hip_lrt2.so DVBS2_HZ2=1
TT=S2_TABLE_B7 tm DVBS2_LIB=$L/libdvbs2_fec_hip_lrt.so DVBS2_V2=1 DVBS2_SOLO=0
TT=S2_TABLE_B4 tm DVBS2_LIB=$L/libdvbs2_fec_hip_lrt.so DVBS2_V2=0 DVBS2_SOLO=0
cat $O/pytest_gs.log $O/b11.log; tail -c 900 $O/bench_awgn.log
