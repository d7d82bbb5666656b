#!/bin/bash
# SQ instruction-mix / stall counters of one kernel inside the bench step: tools/pmc_kernel.sh <kernel-substring> <outdir>
K=$1; OUT=$2; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA" \
           "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python - <<PY
import sqlite3, glob
for f in sorted(glob.glob("$OUT/p*/p_results.db")):
    db = sqlite3.connect(f)
    rows = db.execute("select kernel_name, counter_name, value from counters_collection where kernel_name like '%$K%'").fetchall()
    agg = {}
    for k, c, v in rows:
        a = agg.setdefault(c, [0, 0.0]); a[0] += 1; a[1] += v
    for c, a in agg.items():
        print(f"{c:32s} per-launch {a[1]/a[0]:16.1f}  (n={a[0]})")
PY
