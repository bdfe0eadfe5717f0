#!/bin/bash
# looking for cliffs: ms per DDPM step over 19 clip lengths x 4 batch sizes, three modes, both denoiser architectures
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
{
for p in f16x3 bf16 f32; do python tools/shape_sweep.py $p arch_mdm_l 2>&1 | grep -v amdgpu.ids; done
for p in f16x3 bf16; do python tools/shape_sweep.py $p arch_mdm 2>&1 | grep -v amdgpu.ids; done
} > gpurun_out/r05/shape_sweep_c12.txt 2>&1
cat gpurun_out/r05/shape_sweep_c12.txt | cut -c1-400
