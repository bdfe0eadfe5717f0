#!/bin/bash
# experiment: the residual GEMMs of 64 clips on three-stage 64 x 128 tiles (832 tiles on 512 slots; selection bits 32 + 128) against whole-clip tiles
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
{
for p in bf16 f16x3; do python tools/step_ab.py $p 64 0xa0fffff,-1,0xa0fffff,-1 196 2>&1 | grep -v amdgpu.ids; done
for p in bf16 f16x3; do
  python tools/loop_time.py $p 64 200 3 0xa0fffff 2>&1 | grep ms/step
  python tools/loop_time.py $p 64 200 3 -1 2>&1 | grep ms/step
  python tools/loop_time.py $p 64 200 3 0xa0fffff 2>&1 | grep ms/step
  python tools/loop_time.py $p 64 200 3 -1 2>&1 | grep ms/step
done
} > gpurun_out/r05/b64_resid_deep_c34.txt 2>&1
cut -c1-250 gpurun_out/r05/b64_resid_deep_c34.txt
