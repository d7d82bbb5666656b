#!/bin/bash
# Sample socket power and shader clock beside a probe binary (GPU box): tools/power_trace_probe.sh tools/probes/<name>_probe -> stdout
"$@" > /tmp/probe_out.txt 2>&1 &
BP=$!
for i in $(seq 1 400); do
  echo "t=$SECONDS $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'sclk|Socket' | sed 's/.*: //' | tr '\n' ' ')"
  sleep 0.3
  kill -0 $BP 2>/dev/null || break
done
wait $BP
cat /tmp/probe_out.txt
