#!/bin/bash
# GPU: matrix-pipe-only kernels under the power cap (tools/micro/mfma_power.hip) with rocm-smi sampled beside them:  tools/mfma_power.sh [seconds per case]
secs=${1:-5}
tools/micro/mfma_power $secs > gpurun_out/mfma_power_cases.log 2>&1 &
pid=$!
while kill -0 $pid 2>/dev/null; do
  echo "[t=$(date +%s)] $(rocm-smi --showpower --showclocks 2>/dev/null | grep -E 'Power|sclk' | tr -s ' \t' ' ' | tr '\n' ';')"
  sleep 1
done
wait $pid
cat gpurun_out/mfma_power_cases.log
