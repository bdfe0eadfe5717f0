#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
{ python tools/step_ab.py bf16 32 -1,0x100fffff,-1,0x100fffff 196; python tools/step_ab.py bf16 32 -1,0x2fffff,-1,0x2fffff 160;
  for p in f16x3 bf16 f32; do python tools/loop_time.py $p 64 100 3 -1 160; python tools/loop_time.py $p 64 100 3 -1 196; python tools/loop_time.py $p 32 100 3 -1 160; done; } 2>&1 | grep -E "variant|ms/step" > gpurun_out/r04/t160_defaults_c18.txt
cat gpurun_out/r04/t160_defaults_c18.txt
python -m pytest tests/test_hip_fullsize.py -m gpu -x -q -k "t160 or clip_in or selections" > gpurun_out/r04/gpu_tests_c18.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/gpu_tests_c18.log
grep -E "passed|failed|rc=|Error|assert" gpurun_out/r04/gpu_tests_c18.log | tail -5
