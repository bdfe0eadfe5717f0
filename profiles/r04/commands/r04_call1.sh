#!/bin/bash
# round 4, first GPU call: parity suite on the new status word / weight pre-scaling + baseline loop times of the shapes this round targets
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q > gpurun_out/r04/gpu_tests_c1.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/gpu_tests_c1.log
tail -3 gpurun_out/r04/gpu_tests_c1.log
{
for p in f16x3 bf16 f32; do python tools/loop_time.py $p 64 100 2 -1 196; python tools/loop_time.py $p 64 100 2 -1 160; done
for p in f16x3 bf16; do python tools/loop_time.py $p 32 100 2 -1 196; python tools/loop_time.py $p 32 100 2 -1 160; done
} 2>&1 | grep ms/step > gpurun_out/r04/loop_times_baseline.txt
cat gpurun_out/r04/loop_times_baseline.txt
