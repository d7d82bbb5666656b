#!/bin/bash
# A/B two builds of libsylph_hip.so on the same GPU box: tools/ab.sh <libA> <libB> [bench args]
A=$1; B=$2; shift 2
for r in 1 2; do for L in $A $B; do
  echo "== $(basename $L) run $r: $(SYLPH_LIB_PATH=$PWD/$L timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>&1 | tail -1 | cut -c70-105)"
done; done
