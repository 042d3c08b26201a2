#!/bin/bash
# tools/ab_tables.sh <out> <libA> <libB> tables... -- GPU box: the same tables with two builds of the library, interleaved (A B A B) so that
# clock / thermal drift of the box hits both alike; prints frames/s per table and the ratio B / A
OUT=$1; A=$2; B=$3; shift 3
mkdir -p $(dirname $OUT); : > $OUT
for t in "$@"; do
  for rep in 1 2; do
    for lib in $A $B; do
      f=${lib%%@*}; e=""; [ "$f" != "$lib" ] && e=${lib#*@}   # "lib.so@VAR=value": the same library with an environment override
      fps=$(env $e DVBS2_LIB=$PWD/gr-dvbs2rx_amd/lib/$f timeout 120 python tools/exp_tables.py $t 2>/dev/null | awk '{for(i=1;i<=NF;i++) if($i=="fr/s") print $(i-1)}' | tail -1)
      echo "$t $lib $fps" >> $OUT
    done
  done
done
python - "$OUT" "$A" "$B" <<'PY'
import sys, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for l in open(sys.argv[1]):
    t, lib, f = l.split(); d[t][lib].append(float(f))
for t, v in d.items():
    a, b = max(v[sys.argv[2]]), max(v[sys.argv[3]])
    print(f"{t:24s} A {a/1e3:8.1f} k   B {b/1e3:8.1f} k   B/A {b/a:.3f}")
PY
