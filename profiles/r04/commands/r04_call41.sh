#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
{ python tools/step_ab.py f16x3 64 -1 196; python tools/step_ab.py f16x3 32 -1 196; python tools/step_ab.py f32 32 -1 196; python tools/step_ab.py bf16 32 -1 196; } 2>&1 | grep -v amdgpu > gpurun_out/r04/step_b32_c41.txt
cat gpurun_out/r04/step_b32_c41.txt
