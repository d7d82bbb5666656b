#!/bin/bash
# GPU box: parity of the conv variants, then per-layer A/B of the 3x3 kernels and a short bench
OUT=gpurun_out/r2a; mkdir -p $OUT
timeout 1500 python -m pytest tests/test_conv_variants_gpu.py -x -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log
tail -5 $OUT/tests.log
for cfg in "SYLPH_CONV_HPIPE=1" "SYLPH_CONV_HPIPE=0" "SYLPH_CONV_HPIPE=0 SYLPH_CONV_PATCH_8X16=1"; do
  echo "== $cfg" | tee -a $OUT/layers.log
  env $cfg timeout 300 python tools/bench_layers.py 64 2>&1 | tee -a $OUT/layers.log
done
for cfg in "SYLPH_CONV_HPIPE=1" "SYLPH_CONV_HPIPE=0" "SYLPH_CONV_HPIPE=0 SYLPH_CONV_PATCH_8X16=1"; do
  echo "== $cfg" | tee -a $OUT/bench.log
  env $cfg timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee -a $OUT/bench.log
done
