#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
A=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib/libtamf_hip_A.so
{ echo "A = HEAD before the f32 operand aliasing; default = working tree"
for rep in 1 2; do for cfg in "64 196" "64 160" "32 196"; do set -- $cfg
  TAMF_LIB_OVERRIDE=$A python tools/loop_time.py f32 $1 100 2 -1 $2 2>&1 | grep ms/step
  python tools/loop_time.py f32 $1 100 2 -1 $2 2>&1 | grep ms/step
done; done; } > gpurun_out/r04/ab_f32_alias_c25.txt 2>&1
cat gpurun_out/r04/ab_f32_alias_c25.txt
python -m pytest tests -m gpu -x -q -k "f32" > gpurun_out/r04/gpu_tests_c25.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/gpu_tests_c25.log
grep -E "passed|failed|rc=|Error|assert" gpurun_out/r04/gpu_tests_c25.log | tail -5
