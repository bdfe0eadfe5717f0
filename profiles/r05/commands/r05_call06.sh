#!/bin/bash
# deferred LayerNorm at 32 clips per GPU (BASELINE configs[2] shard) and at T = 160: per-kernel and loops against round 4 (lib A)
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
A=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib/libtamf_hip_A.so
{
for p in f16x3 bf16; do
  echo "== per-kernel, $p, B=32: round-4 library, then working tree"
  TAMF_LIB_OVERRIDE=$A python tools/step_ab.py $p 32 2>&1 | grep -v amdgpu.ids
  python tools/step_ab.py $p 32 2>&1 | grep -v amdgpu.ids
done
echo "== loops B=32, alternating (A = round 4)"
bash tools/ab_loop.sh "f16x3 bf16" 32
echo "== T = 160, B = 64: loops"
for rep in 1 2; do for p in f16x3 bf16; do
  TAMF_LIB_OVERRIDE=$A python tools/loop_time.py $p 64 200 3 -1 160 2>&1 | grep ms/step
  python tools/loop_time.py $p 64 200 3 -1 160 2>&1 | grep ms/step
done; done
} > gpurun_out/r05/ab_deferred_ln_b32_c06.txt 2>&1
cat gpurun_out/r05/ab_deferred_ln_b32_c06.txt | cut -c1-250
