#!/bin/bash
# tools/ab.sh <tables...> -- GPU box: frames/s of each built library variant (gr-dvbs2rx_amd/lib/libdvbs2_fec_hip*.so)
T=${@:-S2_TABLE_B4:50:4096 S2X_TABLE_B21:50:4096}
for so in gr-dvbs2rx_amd/lib/libdvbs2_fec_hip*.so; do
  echo "== $so"
  DVBS2_LIB=$PWD/$so python tools/exp_tables.py $T 2>&1 | grep -v amdgpu
done
