#!/bin/bash
# the same experiment with the variants alternating (the first variant of a process runs ~ 4 % slower than the second: call 32)
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
{
for p in f16x3 bf16; do python tools/step_ab.py $p 32 0x20fffff,-1,0x20fffff,-1,0x20fffff,-1 196 2>&1 | grep -v amdgpu.ids; done
python tools/step_ab.py f16x3 24 0x20fffff,-1,0x20fffff,-1 196 2>&1 | grep -v amdgpu.ids
for p in f16x3 bf16; do
  python tools/loop_time.py $p 32 200 3 0x20fffff 2>&1 | grep ms/step
  python tools/loop_time.py $p 32 200 3 -1 2>&1 | grep ms/step
  python tools/loop_time.py $p 32 200 3 0x20fffff 2>&1 | grep ms/step
  python tools/loop_time.py $p 32 200 3 -1 2>&1 | grep ms/step
done
} > gpurun_out/r05/mid_batch_resid_c33.txt 2>&1
cut -c1-250 gpurun_out/r05/mid_batch_resid_c33.txt
