#!/bin/bash
# tools/ab_bench.sh "<lib A>" "<lib B>" <rounds> <bench args...>: alternating bench.py runs of two library builds on ONE box
# (SYLPH_LIB_PATH selects the build; "" = the product library).  Prints value / ms_per_step per run.
A=$1; B=$2; R=$3; shift 3
for i in $(seq 1 $R); do
  for L in "$A" "$B"; do
    SYLPH_LIB_PATH=$L python bench.py "$@" --no-cpu-baseline --no-sweep --no-parity --no-live-pmc 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('${L:-product}'.split('/')[-1], d['config']['batch_per_gpu'], d['value'], d['ms_per_step'], d['roofline']['frac'], flush=True)"
  done
done
