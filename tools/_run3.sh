cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
CMD="python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-events --no-sweep --no-parity"
for v in "SYLPH_PW_TILE=2" "SYLPH_PW_TILE=1"; do
  tag=$(echo $v | tr '= ' '__')
  OUT=gpurun_out/r3_tl_$tag; mkdir -p $OUT
  env $v timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/trace -o t -- $CMD > $OUT/trace.log 2>&1 || echo "trace failed"
  python tools/rocpd_timeline.py $OUT/trace/t_results.db > gpurun_out/r3_timeline_$tag.txt
  rm -rf $OUT/trace
done
tail -3 gpurun_out/r3_tl_*/trace.log
