#!/bin/bash
# experiment: the residual GEMMs of 24 - 48 clips on 64 x 128 tiles with a three-stage K pipeline (two workgroups per CU; selection bit 32)
# against the row-part / whole-clip tiles (default)
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
{
for B in 24 32 40 48; do
  for p in f16x3 bf16 f32; do python tools/step_ab.py $p $B -1,0x20fffff 196 2>&1 | grep -v amdgpu.ids; done
done
python tools/step_ab.py f16x3 32 -1,0x20fffff 160 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r05/mid_batch_resid_c32.txt 2>&1
cut -c1-250 gpurun_out/r05/mid_batch_resid_c32.txt
