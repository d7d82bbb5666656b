#!/bin/bash
OUT=gpurun_out/r2b; mkdir -p $OUT; rm -f $OUT/*.log
timeout 1500 python -m pytest tests/test_conv_variants_gpu.py -x -q -m gpu > $OUT/tests.log 2>&1; echo "tests rc=$?" >> $OUT/tests.log
tail -4 $OUT/tests.log
V=sylph-few-shot-detection_amd/lib/variants
for r in 1 2; do
for lib in sylph-few-shot-detection_amd/lib/libsylph_hip.so $V/libsylph_noload.so $V/libsylph_nolds.so; do
  SYLPH_LIB_PATH=$PWD/$lib timeout 300 python tools/bench_3x3.py 64 20 2>&1 | tail -1 | tee -a $OUT/ablate.log
done; done
SYLPH_CONV_HPIPE=0 timeout 300 python tools/bench_3x3.py 64 20 2>&1 | tail -1 | tee -a $OUT/ablate.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-200 | tee -a $OUT/bench.log
bash tools/pmc_mfma.sh gpurun_out/r2c 64 tower 2>&1 | tail -2
