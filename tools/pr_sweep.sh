for t in S2_TABLE_B1 S2_TABLE_B2 S2_TABLE_B3 S2_TABLE_B4 S2X_TABLE_B1 S2X_TABLE_B2 S2X_TABLE_B3 S2X_TABLE_B4 S2X_TABLE_B5 S2X_TABLE_B6 S2X_TABLE_C1 S2X_TABLE_C2 S2X_TABLE_C3 S2X_TABLE_C8 S2X_TABLE_C9 S2X_TABLE_C10 S2_TABLE_C1 S2_TABLE_C2 S2_TABLE_C3 S2_TABLE_C4 T2_TABLE_A3; do
  a=$(python tools/exp_tables.py $t:20:4096 2>&1 | tail -1 | awk '{print $2,$3,$4,$5, $8, $9}')
  b=$(DVBS2_PR=1 DVBS2_OCC=1 python tools/exp_tables.py $t:20:4096 2>&1 | grep -v amdgpu | tail -2 | tr '\n' ' ' | awk '{ for(i=1;i<=NF;i++) if ($i=="fr/s") printf "%s ", $(i-1); if ($0 ~ /pr kernel/) printf "PR"; }')
  echo "$t classic: $a | pr: $b"
done
