#!/bin/bash
# determinism soak of the final library (the deep GEMM's prologue wait and clip_tile_of changed this round): same seed -> same bits, loop after loop
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
{
python tools/soak_determinism.py f16x3 64 196 10 1000 2>&1 | tail -1
python tools/soak_determinism.py bf16 64 196 10 1000 2>&1 | tail -1
python tools/soak_determinism.py f32 64 196 4 1000 2>&1 | tail -1
python tools/soak_determinism.py f16x3 1 160 20 1000 2>&1 | tail -1
python tools/soak_determinism.py f32 1 196 10 1000 2>&1 | tail -1
python tools/soak_determinism.py bf16 2 160 20 1000 2>&1 | tail -1
python tools/soak_determinism.py f16x3 5 160 10 1000 2>&1 | tail -1
python tools/soak_determinism.py f16x3 32 196 8 1000 2>&1 | tail -1
} > gpurun_out/r06/soak_determinism_c12.txt 2>&1
cat gpurun_out/r06/soak_determinism_c12.txt
