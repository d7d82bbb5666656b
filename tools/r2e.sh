#!/bin/bash
OUT=gpurun_out/r2e; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events --no-sweep --no-parity"
for f in 1 0; do
  (cd /tmp && SYLPH_GN_FUSE=$f timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$OUT/trace$f -o t -- $CMD > $GRAFT_REPO_ROOT/$OUT/trace$f.log 2>&1)
  python tools/rocpd_timeline.py $(find $OUT/trace$f -name "*_results.db" | head -1) > $OUT/timeline_fuse$f.txt
done
tail -32 $OUT/timeline_fuse1.txt; echo ====; tail -40 $OUT/timeline_fuse0.txt | head -34
