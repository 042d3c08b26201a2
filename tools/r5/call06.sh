#!/bin/bash
# round 5, call 6: cycle stamps of the hazard layers with the packed first / last phase on / off (timing kernel = packed pair build; DVBS2_V2=1
# makes the host lay out packed records for tables whose policy is the plain build)
O=gpurun_out/r5f; mkdir -p $O
for t in S2_TABLE_B11 S2_TABLE_B9; do for v in 1 0; do
  DVBS2_V2=1 DVBS2_V2P=$v DVBS2_TIMING=1 DVBS2_TIMING_LAYERS=1 DVBS2_LIB=$PWD/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip_timing.so timeout 300 python tools/exp_tables.py $t:50:1024 > $O/stamps_${t}_v2p$v.txt 2>&1
  echo "== $t V2P=$v"; grep -E "layer|hazard|timing" $O/stamps_${t}_v2p$v.txt | tail -$(( $(python -c "print({'S2_TABLE_B11':21,'S2_TABLE_B9':33}['$t'])") ))
done; done
