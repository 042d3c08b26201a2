#!/bin/bash
# tools/build_variant.sh <name> "<-D flags>" -- builds gr-dvbs2rx_amd/lib/libdvbs2_fec_hip_<name>.so for kernel A/B experiments
# (select it on the GPU box with DVBS2_LIB=$PWD/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip_<name>.so)
set -e
make -C "$(dirname "$0")/../gr-dvbs2rx_amd" -j8 BUILD=build_$1 OUT=lib/libdvbs2_fec_hip_$1.so EXTRA="$2" 2>&1 | grep -E "error|Error" -A5 || true
ls -la "$(dirname "$0")/../gr-dvbs2rx_amd/lib/libdvbs2_fec_hip_$1.so"
