#!/usr/bin/env python3
"""Per optimizer step of a rocprofv3 rocpd database: interval between successive marker kernels (us) and, inside it, the
summed duration of the kernels whose names contain each of the given substrings.
usage: dbsteps.py DB marker substr [substr ...]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
marker, subs = sys.argv[2], sys.argv[3:]
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
rows = db.execute(f"select d.start, d.end, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
marks = [i for i, r in enumerate(rows) if marker in r[2]]
print("step interval_us " + " ".join(subs))
for n, (a, b) in enumerate(zip(marks, marks[1:])):
    seg = rows[a + 1:b + 1]
    print(n, "%.0f" % ((rows[b][0] - rows[a][0]) / 1e3), " ".join("%.0f" % (sum(e - s for s, e, nm in seg if sub in nm) / 1e3) for sub in subs))
