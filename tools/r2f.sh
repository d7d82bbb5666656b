#!/bin/bash
OUT=gpurun_out/r2f; mkdir -p $OUT; rm -f $OUT/*.log
timeout 900 python -m pytest tests/test_known_answers_gpu.py tests/test_hip_parity.py -m gpu -q -s -k "resnet50 or backbone_fpn_bf16 or full_size_prop or c3_full" 2>&1 | tail -12 | tee $OUT/tests.log
for cfg in "SYLPH_FUSE_BOTTLENECK=2" "SYLPH_FUSE_BOTTLENECK=0" "SYLPH_FUSE_BOTTLENECK=2" "SYLPH_FUSE_BOTTLENECK=1"; do
  echo "== $cfg" | tee -a $OUT/bench.log
  env $cfg timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sweep 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['backbone_fpn'], d.get('parity_bf16'))" | tee -a $OUT/bench.log
done
bash tools/r2e.sh 12
