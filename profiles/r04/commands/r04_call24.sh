#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
python -m pytest tests/test_hip_fullsize.py tests/test_hip_rccl.py -m gpu -x -q -s -k "arch_mdm_b64_t160 or refine_b64_t160 or rccl or presets" > gpurun_out/r04/gpu_tests_c24.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/gpu_tests_c24.log
grep -E "max\|err\||passed|failed|rc=|Error|assert" gpurun_out/r04/gpu_tests_c24.log | tail -12
