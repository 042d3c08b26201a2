#!/bin/bash
# tools/ab3.sh <out> "<lib1 lib2 ...>" tables... -- interleaved runs of several library builds (best of two each)
OUT=$1; LIBS=$2; shift 2
mkdir -p $(dirname $OUT); : > $OUT
for t in "$@"; do for rep in 1 2; do for lib in $LIBS; do
  fps=$(DVBS2_LIB=$PWD/gr-dvbs2rx_amd/lib/$lib timeout 120 python tools/exp_tables.py $t 2>/dev/null | awk '{for(i=1;i<=NF;i++) if($i=="fr/s") print $(i-1)}' | tail -1)
  echo "$t $lib $fps" >> $OUT
done; done; done
python - "$OUT" $LIBS <<'PY'
import sys, collections
d = collections.defaultdict(lambda: collections.defaultdict(list))
for l in open(sys.argv[1]):
    w = l.split()
    if len(w) == 3: d[w[0]][w[1]].append(float(w[2]))
libs = sys.argv[2:]
print(f"{'table':24s} " + " ".join(f"{l.replace('libdvbs2_fec_hip','').replace('.so','') or '(tree)':>12s}" for l in libs))
for t, v in d.items():
    base = max(v[libs[0]])
    print(f"{t:24s} " + " ".join(f"{max(v[l])/1e3:7.1f}k{max(v[l])/base:5.2f}" for l in libs))
PY
