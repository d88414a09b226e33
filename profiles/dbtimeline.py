#!/usr/bin/env python3
"""Timeline of the last optimizer step in a rocprofv3 rocpd database: start / end (us, relative) / queue / kernel.
usage: dbtimeline.py DB [marker-kernel-substring=k_commit] [steps-back=2]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "k_commit"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in db.execute(f"pragma table_info({kd})")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
rows = db.execute(f"select d.start, d.end, {('d.' + qcol) if qcol else '0'}, s.kernel_name from {kd} d join {ks} s on d.kernel_id = s.id order by d.start").fetchall()
marks = [i for i, r in enumerate(rows) if marker in r[3]]
if len(marks) < back + 1:
    sys.exit("not enough steps")
lo, hi = marks[-back - 1] + 1, marks[-back] + 1
t0 = rows[lo][0]
for st, en, q, name in rows[lo:hi]:
    short = name.split("(")[0].replace("_ZN4snes", "").replace("void snes::", "")[:48]
    print("%9.1f %9.1f %8.1f  q%-4s %s" % ((st - t0) / 1e3, (en - t0) / 1e3, (en - st) / 1e3, q, short))
