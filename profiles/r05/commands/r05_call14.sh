#!/bin/bash
# soak: the same seed must give the same bits, loop after loop (the new three-slot statistics staging, the residual prefetch, the
# counted waits of the new epilogues): f16x3 and bf16 at the bench shape, B = 128 (two to four tiles per workgroup: the boundary staging),
# T = 160, and the R-sized architecture (d = 256: two column tiles per clip)
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
{
python tools/soak_determinism.py f16x3 64 196 12 1000 2>&1 | tail -2
python tools/soak_determinism.py bf16 64 196 16 1000 2>&1 | tail -2
python tools/soak_determinism.py f16x3 128 196 6 500 2>&1 | tail -2
python tools/soak_determinism.py bf16 128 160 10 500 2>&1 | tail -2
python tools/soak_determinism.py bf16x3 32 196 10 500 2>&1 | tail -2
python tools/soak_determinism.py f32 64 196 3 500 2>&1 | tail -2
} > gpurun_out/r05/soak_determinism_c14.txt 2>&1
cat gpurun_out/r05/soak_determinism_c14.txt
