#!/bin/bash
# determinism soak at 1e21e24: calls of 1 - 6 clips (32 x 64 tiles with six stages) and 24 - 32 clips (three-stage 64 x 128 tiles)
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
{
python tools/soak_determinism.py f16x3 1 160 20 1000 2>&1 | tail -1
python tools/soak_determinism.py f32 1 196 10 1000 2>&1 | tail -1
python tools/soak_determinism.py bf16 2 160 20 1000 2>&1 | tail -1
python tools/soak_determinism.py f16x3 5 160 10 1000 2>&1 | tail -1
python tools/soak_determinism.py bf16x3 3 123 10 500 2>&1 | tail -1
python tools/soak_determinism.py bf16 32 196 10 1000 2>&1 | tail -1
python tools/soak_determinism.py f16x3 24 196 6 1000 2>&1 | tail -1
python tools/soak_determinism.py f16x3 64 196 6 1000 2>&1 | tail -1
} > gpurun_out/r05/soak_determinism_c39.txt 2>&1
cat gpurun_out/r05/soak_determinism_c39.txt
