#!/bin/bash
# tools/vgprs.sh <DMAX> ["-D flags"] -- registers, spills and scratch of every sweep-kernel build of one degree class
# (device-only compile of csrc/ldpc_inst_<DMAX>.hip, then the code object's metadata notes)
set -e
cd "$(dirname "$0")/../gr-dvbs2rx_amd"
B=/opt/rocm/lib/llvm/bin; T=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off $2 --offload-device-only -c csrc/ldpc_inst_$1.hip -o $T/k.bundle
$B/clang-offload-bundler --unbundle --type=o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --input=$T/k.bundle --output=$T/k.elf
$B/llvm-readelf --notes $T/k.elf | grep -E "\.name:|\.vgpr_count|\.sgpr_spill_count|private_segment_fixed_size|vgpr_spill" | paste - - - - - | \
  sed -E 's/.*\.name: *_ZN5dvbs2[0-9]*([a-z_]+)I([^E]*E[^E]*E[^E]*E[^E]*E[^E]*E[^E]*)E.*private_segment_fixed_size: *([0-9]+).*sgpr_spill_count: *([0-9]+).*vgpr_count: *([0-9]+).*vgpr_spill_count: *([0-9]+)/\1<\2>  scratch \3  sgpr-spills \4  vgprs \5  vgpr-spills \6/'
rm -r "$T"
