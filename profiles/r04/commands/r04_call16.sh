#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
{ for p in f16x3 f32 bf16; do python tools/loop_time.py $p 32 100 2 -1 196; python tools/loop_time.py $p 32 100 2 -1 160; done; python tools/loop_time.py f16x3 64 100 2 -1 196; } 2>&1 | grep ms/step > gpurun_out/r04/loop_times_b32_c16.txt
cat gpurun_out/r04/loop_times_b32_c16.txt
python -m pytest tests/test_hip_fullsize.py tests/test_hip_forward.py -m gpu -x -q > gpurun_out/r04/gpu_tests_c16.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/gpu_tests_c16.log
grep -E "passed|failed|rc=|Error|assert" gpurun_out/r04/gpu_tests_c16.log | tail -5
