#!/bin/bash
# GPU box: randomized parity soak of the round-4 kernels against the genuine reference
O=gpurun_out/r4fuzz; mkdir -p $O
(python tools/fuzz_ldpc.py ${FUZZ_S:-420} 31 2>&1 | tail -3) > $O/fuzz_policy.log &
(DVBS2_GROUP_SPIN_MAX=0 python tools/fuzz_ldpc.py ${FUZZ_S:-420} 32 2>&1 | tail -3) > $O/fuzz_giveup.log &
(DVBS2_GROUP_SYNC=0 python tools/fuzz_ldpc.py ${FUZZ_S:-420} 33 2>&1 | tail -3) > $O/fuzz_nogs.log &
(DVBS2_PR=0 DVBS2_DENSE=0 DVBS2_HZ2=0 DVBS2_V2=0 DVBS2_SOLO=0 python tools/fuzz_ldpc.py ${FUZZ_S:-420} 34 2>&1 | tail -3) > $O/fuzz_plain.log &
wait
python tools/fuzz_bch.py 60 2>&1 | tail -2 > $O/fuzz_bch.log
for f in $O/*.log; do tail -1 $f; done
