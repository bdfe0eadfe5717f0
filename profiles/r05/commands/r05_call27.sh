#!/bin/bash
# residual GEMMs of a few clips on 32- / 64-row tiles with a four-stage K pipeline (tamf_gemm_deep.h): invariance / kernel tests, then per-kernel step profiles with the tiles on (default)
# and off (selection bit 16 = tuning word 0x10fffff) at B = 1 ... 24
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
{
timeout 1500 python -m pytest tests/test_hip_kernels.py tests/test_hip_fullsize.py -x -q -m gpu -k "resid or equals_clip_alone or batch_sizes or same_bits" 2>&1 | tail -4
for B in 1 4 8 12 16 24; do
  for p in f16x3 f32; do python tools/step_ab.py $p $B -1,0x10fffff 160 2>&1 | grep -v amdgpu.ids; done
done
python tools/step_ab.py f16x3 1 -1,0x10fffff 196 2>&1 | grep -v amdgpu.ids
python tools/step_ab.py bf16 1 -1,0x10fffff 160 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r05/small_batch_resid_c27.txt 2>&1
cut -c1-260 gpurun_out/r05/small_batch_resid_c27.txt
