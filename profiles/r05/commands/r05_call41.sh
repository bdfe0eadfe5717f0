#!/bin/bash
# last state of the round (e54fdc3): smoke and the full GPU suite
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
python __graft_entry__.py smoke 2>&1 | tail -4 > gpurun_out/r05/smoke_c41.log
( time timeout 3000 python -m pytest tests -q -m gpu ) > gpurun_out/r05/gpu_tests_full_c41.log 2>&1
cat gpurun_out/r05/smoke_c41.log; tail -6 gpurun_out/r05/gpu_tests_full_c41.log
