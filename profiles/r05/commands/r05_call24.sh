#!/bin/bash
# (1) joules per MAC of the 16x16x32 and 32x32x16 matrix instructions (the split modes run at the power cap: what the step time follows);
# (2) one clip per call: where the 0.75 ms of a DDPM step go, per launch
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
bash tools/mfma_shapes.sh gpurun_out/r05/mfma_shapes_c24.txt 6
{
python tools/step_ab.py f16x3 1 -1 160 2>&1 | grep -v amdgpu.ids
python tools/step_ab.py f16x3 1 -1 196 2>&1 | grep -v amdgpu.ids
python tools/step_ab.py f16x3 8 -1 160 2>&1 | grep -v amdgpu.ids
python tools/step_ab.py f32 1 -1 160 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r05/step_b1_c24.txt 2>&1
cat gpurun_out/r05/step_b1_c24.txt
