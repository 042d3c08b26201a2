#!/bin/bash
# tools/quick8.sh -- experiment helper: rebuild only the DMAX=8 LDPC kernel TU (+ host TUs) and relink.
# The other ldpc_inst_*.o keep their last build; run `make -C gr-dvbs2rx_amd -B -j8` before committing.
set -e
cd "$(dirname "$0")/../gr-dvbs2rx_amd"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-result"
for t in ${@:-ldpc_inst_8}; do /opt/rocm/bin/hipcc $F -c csrc/$t.hip -o build/$t.o & done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libdvbs2_fec_hip.so build/*.o
