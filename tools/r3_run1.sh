#!/bin/bash
# GPU box, round 3 run 1: new parity tests, default bench, cycle stamps of the BASELINE tables
mkdir -p gpurun_out/r3a
timeout 900 python -m pytest tests/test_ldpc_gpu.py tests/test_bch_demap_gpu.py -x -q -m gpu -k "multi_chunk or config1 or full_batch or two_devices or async_entry or enqueue_finish or chain or bch_error or reference_digests" 2>&1 | tail -15 > gpurun_out/r3a/pytest_new.log
timeout 600 python bench.py > gpurun_out/r3a/bench.log 2>&1
for t in S2_TABLE_B4 S2_TABLE_B7 S2_TABLE_B11 S2X_TABLE_B21; do
  echo "== $t" >> gpurun_out/r3a/timing.log
  DVBS2_LIB=$PWD/gr-dvbs2rx_amd/lib/libdvbs2_fec_hip_timing.so DVBS2_TIMING=1 DVBS2_TIMING_LAYERS=1 DVBS2_TIMING_WAVES=1 DVBS2_V2=0 DVBS2_SOLO=0 DVBS2_SOFT_BARRIER=0 timeout 300 python tools/exp_tables.py $t:10:512 2>&1 | grep -v amdgpu.ids | tail -130 >> gpurun_out/r3a/timing.log
done
tail -5 gpurun_out/r3a/pytest_new.log; tail -c 1500 gpurun_out/r3a/bench.log
# operating point: where the time goes (first pass / targets / resume / finalize)
export TMPDIR=/tmp; R=$PWD
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r3a/awgn_trace -o t -- python $R/bench.py --only config2_awgn --no-cpu-baseline --gate none > $R/gpurun_out/r3a/awgn_trace.log 2>&1)
python tools/pmc_summary.py gpurun_out/r3a/awgn_trace/t_results.db > gpurun_out/r3a/awgn_summary.txt 2>&1
rm -rf gpurun_out/r3a/awgn_trace
