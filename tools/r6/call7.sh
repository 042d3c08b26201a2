#!/bin/bash
# GPU box, round 6 call 7: the chunked syndrome pre-test -- bit-exactness, stamps, A/B against the previous library
O=gpurun_out/r6g; mkdir -p $O
timeout 900 python -m pytest tests/test_ldpc_gpu.py -x -q -k "test_every_table_bit_exact and (policy or plain or packed-pair)" -n 4 > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -2 $O/pytest.log
for T in S2_TABLE_B11 S2_TABLE_B7 S2_TABLE_B4; do for L in timing timing0; do
  DVBS2_LIB=$PWD/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip_$L.so DVBS2_TIMING=1 DVBS2_V2=1 WARM_S=0.05 timeout 300 python tools/exp_tables.py $T:10:512 2>&1 | grep "timing," > $O/synd_${T}_$L.txt
  echo "$T $L: $(cat $O/synd_${T}_$L.txt)"
done; done
timeout 2400 python tools/abx.py --out $O/ab.txt --reps 3 --spec "base=libdvbs2_fec_hip_base.so" --spec tree \
  S2_TABLE_B4:50:4096 S2_TABLE_B7:50:4096 S2_TABLE_B11:50:4096 S2X_TABLE_B21:50:4096 S2_TABLE_C1:25:16384 S2_TABLE_B9:50:4096 S2_TABLE_B10:50:4096 S2_TABLE_B8:50:4096 S2_TABLE_B5:50:4096 S2X_TABLE_B10:50:4096 S2_TABLE_B1:50:4096 S2_TABLE_C7:25:8192 S2_TABLE_C5:25:8192
