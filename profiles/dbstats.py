#!/usr/bin/env python3
"""Per-kernel summary (calls, total, average, max) from a rocprofv3 rocpd results database."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start), max(d.end-d.start), max(s.arch_vgpr_count) "
                  f"from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
print("%-70s %7s %12s %12s %12s %5s %6s" % ("kernel", "calls", "total_us", "avg_us", "max_us", "vgpr", "%"))
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    print("%-70s %7d %12.1f %12.1f %12.1f %5d %6.2f" % (r[0][:70], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5], 100.0 * r[2] / tot))
