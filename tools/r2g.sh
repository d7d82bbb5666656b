#!/bin/bash
# A/B timing of bottleneck64_kernel variants (tools/build_variant.sh bk_* ...): kernel-trace durations of the res2 blocks
OUT=gpurun_out/r2g; mkdir -p $OUT; export TMPDIR=/tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events --no-sweep --no-parity"
for v in "" $@; do
  lib=$GRAFT_REPO_ROOT/sylph-few-shot-detection_amd/lib/libsylph_hip.so
  [ -n "$v" ] && lib=$GRAFT_REPO_ROOT/sylph-few-shot-detection_amd/lib/variants/libsylph_$v.so
  rm -rf $OUT/trace_$v
  (cd /tmp && SYLPH_LIB_PATH=$lib timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$OUT/trace_$v -o t -- $CMD > $GRAFT_REPO_ROOT/$OUT/trace_$v.log 2>&1)
  echo "== ${v:-default}"; python tools/rocpd_timeline.py $(find $OUT/trace_$v -name "*_results.db" | head -1) | grep bottleneck64
  rm -rf $OUT/trace_$v
done
