#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
python -m pytest tests/test_hip_kernels.py tests/test_hip_fullsize.py -m gpu -x -q > gpurun_out/r04/gpu_tests_c5.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/gpu_tests_c5.log
grep -E "passed|failed|rc=|Error|assert" gpurun_out/r04/gpu_tests_c5.log | tail -8
{
for p in f16x3 bf16 f32 bf16x3; do python tools/step_ab.py $p 64 -1,0x7ffff; done
for p in f16x3 bf16; do python tools/step_ab.py $p 32 -1,0x7ffff; done
for p in f16x3 bf16 f32; do python tools/loop_time.py $p 64 100 2 -1 196; python tools/loop_time.py $p 64 100 2 0x7ffff 196; done
for p in f16x3 bf16; do python tools/loop_time.py $p 64 100 2 -1 160;  python tools/loop_time.py $p 32 100 2 -1 196; done
} 2>&1 | grep -E "ms/step|variant" > gpurun_out/r04/rowblock_ab_c5.txt
cat gpurun_out/r04/rowblock_ab_c5.txt
