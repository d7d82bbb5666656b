#!/bin/bash
# Memory-path PMC passes for one layer shape: tools/pmc_mem.sh <layer> <outdir>
L=$1; OUT=$2; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -o "TCP_[A-Z0-9_]*" | sort -u | tr '\n' ' ' > $OUT/tcp_counters.txt
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_REQ_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p -- python tools/pmc_layer.py $L > $OUT/p$i.log 2>&1 || echo "pass $i failed/timeout"
done
python - <<PY
import sqlite3, glob
for f in sorted(glob.glob("$OUT/p*/p_results.db")):
    db = sqlite3.connect(f)
    rows = db.execute("select kernel_name, counter_name, value from counters_collection where kernel_name like '%conv_%'").fetchall()
    agg = {}
    for k, c, v in rows:
        a = agg.setdefault((k[:40], c), [0, 0.0]); a[0] += 1; a[1] += v
    for (k, c), a in agg.items():
        print(f"{k:40s} {c:40s} per-launch {a[1]/a[0]:16.1f}  (n={a[0]})")
PY
