#!/bin/bash
# Sample socket power and shader clock beside a bench run (GPU box): tools/power_trace.sh [steps]  -> stdout
python bench.py --steps ${1:-600} --no-sweep --no-parity --no-cpu-baseline > gpurun_out/power_bench.json 2>/dev/null &
BP=$!
for i in $(seq 1 200); do
  echo "t=$SECONDS $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'sclk|Socket' | sed 's/.*: //' | tr '\n' ' ')"
  sleep 0.4
  kill -0 $BP 2>/dev/null || break
done
wait $BP
tail -c 400 gpurun_out/power_bench.json
