#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
python -m pytest tests/test_hip_kernels.py -m gpu -x -q -k "ln" > gpurun_out/r04/gpu_tests_c6.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/gpu_tests_c6.log
grep -E "passed|failed|rc=|Error|assert" gpurun_out/r04/gpu_tests_c6.log | tail -5
{
for p in f16x3 bf16 f32; do python tools/step_ab.py $p 64 -1,0x7ffff; done
for p in f16x3 bf16; do python tools/step_ab.py $p 32 -1,0x7ffff; done
} 2>&1 | grep -E "ms/step|variant" > gpurun_out/r04/rowblock_ab_c6.txt
cat gpurun_out/r04/rowblock_ab_c6.txt
