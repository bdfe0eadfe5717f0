#!/bin/bash
# GPU: sample rocm-smi (power, clocks, temperature) while the 1000-step loop runs:  tools/power_trace.sh [dtype] [seconds]
dt=${1:-f16x3}; secs=${2:-12}
python3 tools/loop_time.py $dt 64 1000 8 > gpurun_out/power_loop_$dt.log 2>&1 &
pid=$!
sleep 6   # context set-up, graph capture
for i in $(seq 1 $secs); do
  rocm-smi --showpower --showclocks --showtemp --showuse 2>/dev/null | grep -E "Power|sclk|mclk|fclk|Temperature \(Sensor (junction|edge)|GPU use" | tr -s ' ' | tr '\n' ';'
  echo
  sleep 1
done
wait $pid
tail -3 gpurun_out/power_loop_$dt.log
