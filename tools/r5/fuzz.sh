#!/bin/bash
# GPU box: randomized parity soak of the round-5 kernels against the genuine reference (FUZZ_S seconds each, four legs in parallel)
O=gpurun_out/${1:-r5fuzz}; mkdir -p $O
(python tools/fuzz_ldpc.py ${FUZZ_S:-300} 51 2>&1 | tail -3) > $O/fuzz_policy.log &
(DVBS2_GROUP_SPIN_MAX=0 python tools/fuzz_ldpc.py ${FUZZ_S:-300} 52 2>&1 | tail -3) > $O/fuzz_giveup.log &
(DVBS2_PR=0 DVBS2_DENSE=0 DVBS2_HZ2=0 DVBS2_V2=1 DVBS2_SOLO=0 python tools/fuzz_ldpc.py ${FUZZ_S:-300} 53 2>&1 | tail -3) > $O/fuzz_packed_pair.log &
(DVBS2_PR=0 DVBS2_DENSE=0 DVBS2_HZ2=0 DVBS2_V2=0 DVBS2_SOLO=0 python tools/fuzz_ldpc.py ${FUZZ_S:-300} 54 2>&1 | tail -3) > $O/fuzz_plain.log &
wait
python tools/fuzz_bch.py 45 2>&1 | tail -2 > $O/fuzz_bch.log
DVBS2_BCH_SYND_MIN=1 python tools/fuzz_bch.py 45 2>&1 | tail -2 > $O/fuzz_bch_product.log   # syndromes through the batched matrix product
for f in $O/*.log; do echo "$f: $(tail -1 $f)"; done
