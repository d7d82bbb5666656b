#!/usr/bin/env python
"""Per-kernel LDS bank-conflict share from a rocprofv3 PMC pass (rocpd db):

    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d out -o l -- python bench.py ...
    python tools/rocpd_lds_conflicts.py out/l_results.db

SQ_LDS_BANK_CONFLICT = LDS-array cycles added by conflicts, SQ_LDS_IDX_ACTIVE = all LDS-array cycles (MI355X_MICROARCH.md, LDS section):
their ratio is the share of a kernel's LDS time that a better layout would remove."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
agg = {}
for name, counter, value in db.execute("select kernel_name, counter_name, value from counters_collection"):
    k = name.split("(")[0].replace("sylph::", "").replace("void ", "")[:64]
    d = agg.setdefault(k, {"n": 0})
    d[counter] = d.get(counter, 0.0) + value
    if counter == "SQ_LDS_IDX_ACTIVE":
        d["n"] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_LDS_IDX_ACTIVE", 0.0))
print(f"{'kernel':64s} {'launches':>8s} {'LDS cycles':>14s} {'conflict cycles':>16s} {'share':>6s}")
for k, d in rows[:24]:
    act, conf = d.get("SQ_LDS_IDX_ACTIVE", 0.0), d.get("SQ_LDS_BANK_CONFLICT", 0.0)
    if act > 0:
        print(f"{k:64s} {d['n']:8d} {act:14.0f} {conf:16.0f} {conf / act:6.2f}")
