#!/bin/bash
# flakiness check of the two-thread test (20 repetitions) and of the stub-CLIP / resize tests
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
for i in $(seq 1 20); do
  timeout 300 python -m pytest tests/test_hip_robustness.py -q -m gpu -k "two_threads or two_contexts" 2>&1 | tail -1
done > gpurun_out/r06/two_threads_repeat_c14.txt 2>&1
sort gpurun_out/r06/two_threads_repeat_c14.txt | uniq -c
