#!/usr/bin/env python3
"""Per-kernel call counts and mean durations from a rocprofv3 rocpd database (gpurun_out/<dir>/**/*.db).

    rocprofv3 --kernel-trace -d gpurun_out/<dir> -- python3 bench.py ...
    python tools/kernel_stats.py gpurun_out/<dir>
"""
import glob
import sqlite3
import sys

db = sorted(glob.glob(sys.argv[1] + "/**/*.db", recursive=True))[-1]
c = sqlite3.connect(db)
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
q = (f"select s.kernel_name, count(*), avg(d.end-d.start), sum(d.end-d.start) from {kd} d "
     f"join {ks} s on d.kernel_id=s.id group by 1 order by 4 desc limit {int(sys.argv[2]) if len(sys.argv) > 2 else 14}")
tot = c.execute(f"select sum(end-start) from {kd}").fetchone()[0]
for name, n, avg, s in c.execute(q):
    print(f"{name[:78]:78s} n={n:6d} avg {avg / 1e3:8.2f} us  total {s / 1e6:8.2f} ms  {100 * s / tot:5.1f}%")
