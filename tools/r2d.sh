#!/bin/bash
OUT=gpurun_out/r2d; mkdir -p $OUT; rm -f $OUT/*.log
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -s -k "head_bf16 or c3_full or full_size_prop" 2>&1 | tail -8 | tee $OUT/tests.log
SYLPH_CONV_HPIPE=2 timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -s -k "head_bf16" 2>&1 | tail -4 | tee -a $OUT/tests.log
SYLPH_CONV_HPIPE=2 SYLPH_GN_FUSE=0 timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q -s -k "head_bf16" 2>&1 | tail -4 | tee -a $OUT/tests.log
for r in 1 2; do timeout 300 python tools/bench_3x3.py 64 20 2>&1 | tail -1 | tee -a $OUT/layers.log; done
for cfg in "SYLPH_GN_FUSE=1" "SYLPH_GN_FUSE=0" "SYLPH_GN_FUSE=1" "SYLPH_GN_FUSE=0"; do
  echo "== $cfg" | tee -a $OUT/bench.log
  env $cfg timeout 300 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-sweep 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['head'], d.get('parity_bf16'))" | tee -a $OUT/bench.log
done
