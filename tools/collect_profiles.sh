#!/bin/bash
# Round profile collection on the GPU box (run through gpurun): tools/collect_profiles.sh <tag> [batch = 192]
# kernel-trace summary + separate PMC passes (FETCH_SIZE, WRITE_SIZE) of the default bench command.
TAG=${1:-r6}; OUT=gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
BATCH=${2:-192}
CMD="python bench.py --batch $BATCH --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events --no-sweep --no-parity --no-live-pmc"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1 || echo "trace failed"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o f -- $CMD > $OUT/fetch.log 2>&1 || echo "fetch pass failed"
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o w -- $CMD > $OUT/write.log 2>&1 || echo "write pass failed"
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $OUT/lds -o l -- $CMD > $OUT/lds.log 2>&1 || echo "lds pass failed"
python tools/rocpd_lds_conflicts.py $OUT/lds/l_results.db > $OUT/lds_bank_conflicts.txt
python tools/rocpd_summary.py $OUT/trace/t_results.db $OUT/kernel_summary.md > /dev/null
python tools/rocpd_timeline.py $OUT/trace/t_results.db > $OUT/timeline.txt
python tools/rocpd_pmc.py $OUT/fetch/f_results.db $OUT/write/w_results.db $BATCH $OUT/pmc_hbm_traffic.json | tail -12
head -30 $OUT/kernel_summary.md
# batch-1 timeline (the serving shape of SylphPredictor / the reference's query loop)
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace_b1 -o t -- python bench.py --batch 1 --steps 20 --warmup 5 --inflight 1 --no-cpu-baseline --no-kernel-events --no-sweep --no-parity --no-live-pmc > $OUT/trace_b1.log 2>&1 || echo "b1 trace failed"
python tools/rocpd_timeline.py $OUT/trace_b1/t_results.db > $OUT/timeline_b1.txt
# the raw rocpd databases stay on the box (gpurun_out/ is capped at 64 MiB): only the summaries travel back
rm -rf $OUT/trace $OUT/fetch $OUT/write $OUT/lds $OUT/trace_b1
