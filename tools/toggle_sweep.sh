#!/bin/bash
# tools/toggle_sweep.sh <tables...> -- GPU box: frames/s per kernel build: packed nodes (DVBS2_V2, with DVBS2_CHAIN_V2) x one-frame workgroups (DVBS2_SOLO)
for t in "$@"; do
  for cfg in "1 1" "0 1" "1 0" "0 0"; do
    set -- $cfg
    r=$(DVBS2_V2=$1 DVBS2_CHAIN_V2=$1 DVBS2_SOLO=$2 WARM_S=0.3 timeout 120 python tools/exp_tables.py $t 2>&1 | grep "fr/s" | awk '{for(i=1;i<=NF;i++) if($i=="fr/s") print $(i-1)}')
    echo "$t packed=$1 solo=$2 : $r fr/s"
  done
done
