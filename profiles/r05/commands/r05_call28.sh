#!/bin/bash
# every GEMM of the step on 32- / 64-row tiles with the four-stage K pipeline when the call holds a few clips: parity / invariance tests,
# per-kernel step profiles with the tiles on (default) and off (tuning word 0x10fffff) at B = 1 ... 24
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
{
timeout 1500 python -m pytest tests/test_hip_kernels.py tests/test_hip_forward.py tests/test_hip_robustness.py tests/test_hip_module.py -x -q -m gpu 2>&1 | tail -3
timeout 1500 python -m pytest tests/test_hip_fullsize.py -x -q -m gpu -k "resid or equals_clip_alone or batch_sizes or same_bits or other_batch" 2>&1 | tail -3
for B in 1 2 4 8 12 16 24; do
  for p in f16x3 f32; do python tools/step_ab.py $p $B -1,0x10fffff 160 2>&1 | grep -v amdgpu.ids; done
done
python tools/step_ab.py f16x3 1 -1,0x10fffff 196 2>&1 | grep -v amdgpu.ids
python tools/step_ab.py bf16 1 -1,0x10fffff 160 2>&1 | grep -v amdgpu.ids
python tools/step_ab.py bf16 8 -1,0x10fffff 160 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r05/small_batch_all_c28.txt 2>&1
cut -c1-260 gpurun_out/r05/small_batch_all_c28.txt
