#!/bin/bash
O=gpurun_out/r3d; mkdir -p $O; L=$PWD/gr-dvbs2rx_amd/lib
tm() { echo "== $TT $*" >> $O/timing.log; env "$@" DVBS2_LIB=$L/libdvbs2_fec_hip_lrt.so DVBS2_TIMING=1 DVBS2_TIMING_LAYERS=1 timeout 300 python tools/exp_tables.py $TT:10:512 2>&1 | grep -v amdgpu.ids | grep -E "hazard phases|fr/s|\[timing" | tail -5 >> $O/timing.log; }
TT=S2_TABLE_B4 tm DVBS2_V2=0 DVBS2_SOLO=0
TT=S2_TABLE_B4 tm DVBS2_V2=0 DVBS2_SOLO=0 DVBS2_LANE_CHAIN_MAX=0
TT=S2_TABLE_B7 tm DVBS2_V2=0 DVBS2_SOLO=0
TT=S2_TABLE_B7 tm DVBS2_V2=0 DVBS2_SOLO=0 DVBS2_LANE_CHAIN_MAX=0
TT=S2_TABLE_B11 tm DVBS2_V2=0
cat $O/timing.log
