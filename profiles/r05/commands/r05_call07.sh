#!/bin/bash
# the full GPU suite at the deferred-LayerNorm state (3fd4e27 + script fixes)
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
( time timeout 3000 python -m pytest tests -q -m gpu ) > gpurun_out/r05/gpu_tests_full_c07.log 2>&1
tail -15 gpurun_out/r05/gpu_tests_full_c07.log
