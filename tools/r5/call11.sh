#!/bin/bash
# round 5, call 11: does S2X 154/180 wait for its bytes? counter traffic of the tree and of the half-the-message-words build (timing-only) beside their rates
O=$PWD/gpurun_out/r5k; mkdir -p $O; export TMPDIR=/tmp; R=$PWD
cd /tmp
for lib in tree oneword; do
  e=""; [ $lib != tree ] && e="DVBS2_LIB=$R/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip_$lib.so"
  for c in FETCH_SIZE WRITE_SIZE; do
    env $e WARM_S=0.2 rocprofv3 --pmc $c -d $O/pmc_${lib}_$c -o p -- python $R/tools/exp_tables.py S2X_TABLE_B21:50:4096 > $O/pmc_${lib}_$c.log 2>&1
  done
done
cd $R
python tools/pmc_summary.py $O/pmc_*/p_results.db 2>&1 | grep -E "^==|layered_kernel" | cut -c1-400 | tee $O/summary.txt
rm -rf $O/pmc_*/
python tools/abx.py --out $O/tree.txt --spec tree S2_TABLE_B11:50:4096 S2_TABLE_B10:50:4096 S2_TABLE_B9:50:4096 S2_TABLE_B8:50:4096 S2X_TABLE_B21:50:4096 S2_TABLE_B7:50:4096 S2_TABLE_B4:50:4096
