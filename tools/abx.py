#!/usr/bin/env python3
"""tools/abx.py -- GPU box: the same tables with several (library build, environment) variants, interleaved so that clock / thermal
drift hits all alike; best of --reps per variant.

  python tools/abx.py --out gpurun_out/x/ab.txt --spec tree --spec "ow=libdvbs2_fec_hip_oneword.so" \
         --spec "v2=,DVBS2_V2=1" S2_TABLE_B4:50:4096 ...

A spec is  name[=lib.so][,VAR=value ...]  (lib relative to gr-dvbs2rx_amd/lib; empty = the tree's library).
"""
import argparse
import collections
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", required=True)
    ap.add_argument("--spec", action="append", required=True)
    ap.add_argument("--reps", type=int, default=2)
    ap.add_argument("tables", nargs="+")
    a = ap.parse_args()
    specs = []
    for s in a.spec:
        parts = s.split(",")
        name, _, lib = parts[0].partition("=")
        env = dict(p.split("=", 1) for p in parts[1:])
        if lib:
            env["DVBS2_LIB"] = os.path.join(ROOT, "gr-dvbs2rx_amd", "lib", lib)
        specs.append((name, env))
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    res = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(a.out, "w") as log:
        for t in a.tables:
            for _ in range(a.reps):
                for name, env in specs:
                    e = dict(os.environ); e.update(env)
                    try:
                        o = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "exp_tables.py"), t], env=e, capture_output=True,
                                           text=True, timeout=180).stdout
                    except subprocess.TimeoutExpired:
                        o = ""
                    m = re.search(r"([0-9.]+) fr/s", o)
                    fps = float(m.group(1)) if m else 0.0
                    res[t][name].append(fps)
                    log.write(f"{t} {name} {fps}\n"); log.flush()
        names = [n for n, _ in specs]
        lines = [f"{'table':28s} " + " ".join(f"{n:>16s}" for n in names)]
        for t in a.tables:
            base = max(res[t][names[0]]) or 1.0
            lines.append(f"{t:28s} " + " ".join(f"{max(res[t][n]) / 1e3:9.1f}k {max(res[t][n]) / base:5.3f}" for n in names))
        log.write("\n".join(lines) + "\n")
        print("\n".join(lines))


if __name__ == "__main__":
    main()
