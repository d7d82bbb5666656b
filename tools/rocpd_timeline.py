#!/usr/bin/env python
"""Print the per-launch timeline of the last query step (from the last preprocess kernel on) of a
rocprofv3 rocpd database.  Usage: python tools/rocpd_timeline.py <results.db>"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
idx = [i for i, r in enumerate(rows) if "preprocess" in r[0]]
if not idx:  # the normalisation is fused into the stem kernel: a step starts with it
    idx = [i for i, r in enumerate(rows) if "stem_pool" in r[0]]
step = rows[idx[-1]:]
t0 = step[0][1]
for n, s, e, g, wg in step:
    if "at::native" in n or "rocclr" in n:
        continue
    nm = re.sub(r"^_ZN5sylph\d+", "", n)
    m = re.search(r"Li(\d+)ELi(\d+)ELi\dELi\dELi\d", nm)
    tag = f"conv{m.group(1)}x{m.group(2)}" if m else re.sub(r"(I|\().*", "", nm.replace("sylph::", "").replace("void ", ""))[:28]
    print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f}us blocks={g // max(wg, 1):6d} {tag}")
