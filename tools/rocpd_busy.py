#!/usr/bin/env python
"""GPU busy fraction of a rocprofv3 rocpd kernel trace: summed kernel durations / span, gaps between consecutive kernels.
Usage: python tools/rocpd_busy.py results.db [skip_first_n_kernels]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else len(rows) // 2
rows = rows[skip:]
span = rows[-1][2] - rows[0][1]
busy = sum(e - s for _, s, e in rows)
gaps = [rows[i + 1][1] - rows[i][2] for i in range(len(rows) - 1)]
pos = sorted(g for g in gaps if g > 0)
print(f"kernels {len(rows)}  span {span / 1e6:.3f} ms  busy {busy / 1e6:.3f} ms ({busy / span:.3f})")
print(f"gaps: n={len(pos)} sum {sum(pos) / 1e6:.3f} ms  median {pos[len(pos) // 2] / 1e3:.2f} us  p90 {pos[int(len(pos) * 0.9)] / 1e3:.2f} us  max {pos[-1] / 1e3:.1f} us")
big = sorted(((rows[i + 1][1] - rows[i][2]), rows[i][0][:50], rows[i + 1][0][:50]) for i in range(len(rows) - 1))[-8:]
for g, a, b in big:
    print(f"  {g / 1e3:8.1f} us between {a} -> {b}")
