#!/bin/bash
# the GPU tests of respaced sampling (fused loop with a timestep map, module contract, error paths) and of cond_fn guidance
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 900 python -m pytest tests/ -q -m gpu -s -k "respaced or timestep_map or cond_fn" 2>&1 | grep -v amdgpu.ids | tail -25 > gpurun_out/r06/gpu_tests_respaced_c11.log
cat gpurun_out/r06/gpu_tests_respaced_c11.log
