#!/bin/bash
# Sample the shader clock and power while a workload runs (GPU box): tools/clock_probe.sh <python args...>
python "$@" > /tmp/clock_probe_workload.log 2>&1 &
PID=$!
sleep 4
for i in 1 2 3 4 5 6; do
  rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk" | tr '\n' ' '
  echo
  sleep 0.5
done
wait $PID
tail -2 /tmp/clock_probe_workload.log
