#!/bin/bash
OUT=gpurun_out/r2e; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events --no-sweep --no-parity"
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$OUT/trace -o t -- $CMD > $GRAFT_REPO_ROOT/$OUT/trace.log 2>&1)
python tools/rocpd_timeline.py $(find $OUT/trace -name "*_results.db" | head -1) > $OUT/timeline.txt
head -${1:-30} $OUT/timeline.txt
