#!/usr/bin/env python3
"""tools/kstats.py <rocprofv3 results .db> -- per-kernel count / average / min / max duration (us) of a --kernel-trace run."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
for r in c.execute("select name, count(*), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 2*avg(end-start) desc").fetchall()[:12]:
    print(f"{r[0][:90]:90s} calls {r[1]:5d}  avg {r[2] / 1e3:9.1f} us  min {r[3] / 1e3:9.1f}  max {r[4] / 1e3:9.1f}")
