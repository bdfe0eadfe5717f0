#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 300 tools/micro/lds_fill > gpurun_out/r04/lds_fill.txt 2>&1
cat gpurun_out/r04/lds_fill.txt
python -m pytest tests -m gpu -x -q > gpurun_out/r04/gpu_tests_c3.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/gpu_tests_c3.log
grep -E "passed|failed|rc=|Error|assert|stress" gpurun_out/r04/gpu_tests_c3.log | tail -12
python -m pytest tests/test_hip_forward.py -m gpu -q -s -k stress 2>&1 | grep "stress" > gpurun_out/r04/stress_errors.txt
cat gpurun_out/r04/stress_errors.txt
