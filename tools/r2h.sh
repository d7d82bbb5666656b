#!/bin/bash
# pointwise-conv tile / staging A-B on the bottleneck layer shapes (tools/bench_layers.py) at B = 64
for cfg in "X=0" "SYLPH_CONV_FORCE_BM=256" "SYLPH_CONV_NBUF=2" "SYLPH_CONV_NBUF=2 SYLPH_CONV_FORCE_BM=256" "SYLPH_CONV_FORCE_BN=256"; do
  echo "== $cfg"; env $cfg timeout 300 python tools/bench_layers.py 64 2>&1 | grep -E "1x1"
done
