mkdir -p gpurun_out
{
python tools/exp_tables.py S2_TABLE_B4:50:512 S2_TABLE_B4:50:4096 2>&1 | grep -v amdgpu
for nf in 512 4096; do for mode in "" pinned; do for ch in "" 128 256; do
  echo -n "nf=$nf mode=${mode:-pageable} chunk=${ch:-default}: "
  env ${ch:+DVBS2_HOST_CHUNK=$ch} python tools/host_api_rate.py $nf $mode 2>&1 | grep -v amdgpu
done; done; done
} > gpurun_out/hostrate.log 2>&1
cat gpurun_out/hostrate.log
