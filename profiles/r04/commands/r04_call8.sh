#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
{
for p in f16x3 bf16 f32; do
  python tools/kbench_one.py $p 2 -1 13312 512 2048 20 2>&1 | tail -1; python tools/kbench_one.py $p 2 -1 13312 512 512 20 2>&1 | tail -1
  python tools/kbench_one.py $p 2 0x7ffff 13312 512 2048 20 2>&1 | tail -1; python tools/kbench_one.py $p 2 0x7ffff 13312 512 512 20 2>&1 | tail -1
done
for p in f16x3 bf16; do python tools/kbench_one.py $p 2 -1 6656 512 2048 20 2>&1 | tail -1; python tools/kbench_one.py $p 2 -1 10752 512 2048 20 2>&1 | tail -1; done
for p in f16x3 bf16 f32; do python tools/step_ab.py $p 64 -1,0x7ffff; done
for p in f16x3 bf16; do python tools/step_ab.py $p 32 -1,0x7ffff; done
} 2>&1 | grep -E "us |variant" > gpurun_out/r04/rowblock_c8.txt
cat gpurun_out/r04/rowblock_c8.txt
python -m pytest tests/test_hip_kernels.py tests/test_hip_fullsize.py -m gpu -x -q -k "ln or selections or clip_in or forward_b64" > gpurun_out/r04/gpu_tests_c8.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/gpu_tests_c8.log
grep -E "passed|failed|rc=|Error|assert" gpurun_out/r04/gpu_tests_c8.log | tail -8
