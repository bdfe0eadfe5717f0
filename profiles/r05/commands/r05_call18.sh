#!/bin/bash
# the full GPU suite at 758471d (every mode on the deferred LayerNorm, legacy LayerNorm kernels removed)
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
( time timeout 3000 python -m pytest tests -q -m gpu ) > gpurun_out/r05/gpu_tests_full_c18.log 2>&1
tail -15 gpurun_out/r05/gpu_tests_full_c18.log
