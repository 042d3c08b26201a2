#!/usr/bin/env python3
"""Summarise rocprofv3 rocpd .db outputs: per kernel, sum of each counter over dispatches (+ durations)."""
import sqlite3, sys, collections
for db in sys.argv[1:]:
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    print("==", db)
    if 'counters_collection' in tabs:
        rows = c.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id), sum(duration)/count(*)*count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name").fetchall()
        agg = collections.defaultdict(dict)
        for k, n, v, nd, dur in rows:
            agg[k][n] = v; agg[k]['_dispatches'] = nd; agg[k]['_dur_ns'] = dur
        for k, d in agg.items():
            if 'ldpc' not in k and 'bch' not in k and 'demap' not in k: continue
            print(k[:110], {a: (round(b) if isinstance(b, float) else b) for a, b in sorted(d.items())})
    if 'kernels' in tabs:
        for r in c.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) from kernels group by name order by 3 desc limit 24"):
            print("%-110s calls %4d total %10.3f ms avg %9.3f ms min %9.3f max %9.3f" % (r[0][:110], r[1], r[2]/1e6, r[3]/1e6, r[4]/1e6, r[5]/1e6))
        # the launches that decode a whole batch (resume launches that find nothing to do and the 32-frame parity gates pull the
        # plain average down): every duration above 1 ms, per kernel
        for (name,) in c.execute("select distinct name from kernels where name like '%ldpc_layered%'").fetchall():
            d = [r[0] / 1e6 for r in c.execute("select end-start from kernels where name=? and end-start > 1000000 order by start", (name,))]
            print("   launches > 1 ms of %s: %s" % (name[:80], " ".join("%.2f" % x for x in d)))
