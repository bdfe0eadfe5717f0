#!/bin/bash
# deferred LayerNorm, first contact: smoke, the parity suites (forward / fullsize / module / robustness), then loop times per mode
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
python __graft_entry__.py smoke > gpurun_out/r05/smoke_c02.log 2>&1; tail -5 gpurun_out/r05/smoke_c02.log
timeout 1500 python -m pytest tests/test_hip_forward.py tests/test_hip_kernels.py -x -q -m gpu > gpurun_out/r05/gpu_tests_forward_c02.log 2>&1; tail -15 gpurun_out/r05/gpu_tests_forward_c02.log
timeout 1800 python -m pytest tests/test_hip_fullsize.py -q -m gpu > gpurun_out/r05/gpu_tests_fullsize_c02.log 2>&1; tail -40 gpurun_out/r05/gpu_tests_fullsize_c02.log
for p in f16x3 bf16 bf16x3; do python tools/loop_time.py $p 64 2>&1 | tail -2; done > gpurun_out/r05/loop_times_c02.txt 2>&1; cat gpurun_out/r05/loop_times_c02.txt
