#!/bin/bash
# determinism soak at bd2c078: the bench shapes again (deferred LayerNorm in f32 too) and calls of a few clips (small tiles with the deep K pipeline)
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
{
python tools/soak_determinism.py f16x3 64 196 8 1000 2>&1 | tail -1
python tools/soak_determinism.py f32 64 196 3 500 2>&1 | tail -1
python tools/soak_determinism.py bf16 64 196 8 1000 2>&1 | tail -1
python tools/soak_determinism.py f16x3 1 160 20 1000 2>&1 | tail -1
python tools/soak_determinism.py f16x3 1 196 10 1000 2>&1 | tail -1
python tools/soak_determinism.py f32 1 160 10 1000 2>&1 | tail -1
python tools/soak_determinism.py bf16 1 160 20 1000 2>&1 | tail -1
python tools/soak_determinism.py f16x3 8 160 10 1000 2>&1 | tail -1
python tools/soak_determinism.py bf16x3 4 196 10 500 2>&1 | tail -1
python tools/soak_determinism.py f32 16 160 4 500 2>&1 | tail -1
} > gpurun_out/r05/soak_determinism_c30.txt 2>&1
cat gpurun_out/r05/soak_determinism_c30.txt
