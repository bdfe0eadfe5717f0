#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
{ python tools/soak_determinism.py f16x3 64 196 12; python tools/soak_determinism.py f16x3 64 160 12; python tools/soak_determinism.py f16x3 32 160 12; python tools/soak_determinism.py f16x3 32 196 12;
  python tools/soak_determinism.py bf16 64 160 12; python tools/soak_determinism.py bf16 32 196 12; python tools/soak_determinism.py f32 64 160 4; python tools/soak_determinism.py f32 32 196 4; python tools/soak_determinism.py bf16x3 64 160 6; } 2>&1 | grep -E "soak|DIFFERENT" > gpurun_out/r04/soak_determinism_c27.txt
cat gpurun_out/r04/soak_determinism_c27.txt
