#!/bin/bash
# Round profile collection on the GPU box (run through gpurun): tools/collect_profiles.sh <tag>
# kernel-trace summary + separate PMC passes (FETCH_SIZE, WRITE_SIZE) of the default bench command.
TAG=${1:-r1_d}; OUT=gpurun_out/$TAG; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events --no-sweep --no-parity"
timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1 || echo "trace failed"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o f -- $CMD > $OUT/fetch.log 2>&1 || echo "fetch pass failed"
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o w -- $CMD > $OUT/write.log 2>&1 || echo "write pass failed"
python tools/rocpd_summary.py $OUT/trace/t_results.db $OUT/kernel_summary.md > /dev/null
python tools/rocpd_timeline.py $OUT/trace/t_results.db > $OUT/timeline.txt
python tools/rocpd_pmc.py $OUT/fetch/f_results.db $OUT/write/w_results.db 64 $OUT/pmc_hbm_traffic.json | tail -12
head -30 $OUT/kernel_summary.md
