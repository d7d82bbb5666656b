#!/usr/bin/env python
"""Per-kernel totals from a rocprofv3 rocpd database (the default output format of this image's rocprofv3 when no
--output-format is given).  Usage: python tools/rocpd_stats.py results.db [steps]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
cols = [r[1] for r in cur.execute(f"pragma table_info({ks})")]
namecol = "kernel_name" if "kernel_name" in cols else "display_name"
rows = list(cur.execute(f"select s.{namecol}, count(*), sum(d.end-d.start), avg(d.end-d.start) from {kd} d join {ks} s "
                        f"on d.kernel_id=s.id group by 1 order by 3 desc"))
tot = sum(r[2] for r in rows)
print(f"{'kernel':64s} {'calls':>7s} {'us/step':>10s} {'avg us':>10s} {'%':>6s}")
for n, c, t, a in rows[:30]:
    name = re.sub(r"\(.*", "", n)[:64]
    print(f"{name:64s} {c:7d} {t / steps / 1e3:10.1f} {a / 1e3:10.1f} {100 * t / tot:6.1f}")
print(f"total {tot / steps / 1e6:.3f} ms/step over {steps} steps")
