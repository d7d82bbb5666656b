#!/bin/bash
# rocprofv3 PMC passes (counters only, kernel-trace) for one layer: tools/pmc_passes.sh <layer> <outdir>
L=$1; OUT=$2; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
           "SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_CYCLES" \
           "TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum" \
           "TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_CYCLE_sum TCC_IB_STALL_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace -d $OUT/p$i -o p -- python tools/pmc_layer.py $L > $OUT/p$i.log 2>&1
done
python - <<PY
import sqlite3, glob
for f in sorted(glob.glob("$OUT/p*/p_results.db")):
    db = sqlite3.connect(f)
    rows = db.execute("select kernel_name, counter_name, value from counters_collection where kernel_name like '%conv_igemm%'").fetchall()
    agg = {}
    for k, c, v in rows:
        a = agg.setdefault(c, [0, 0.0]); a[0] += 1; a[1] += v
    for c, a in agg.items():
        print(f"{c:45s} per-launch {a[1]/a[0]:16.1f}  (n={a[0]})")
PY
