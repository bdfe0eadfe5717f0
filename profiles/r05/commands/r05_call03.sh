#!/bin/bash
# deferred LayerNorm: same-box A/B against the round-4 library (lib A = dd0fa1f) - whole loop and per-kernel HIP-event profile
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
A=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib/libtamf_hip_A.so
{
for p in bf16 f16x3; do
  echo "== per-kernel, $p, B=64: round-4 library, then working tree"
  TAMF_LIB_OVERRIDE=$A python tools/step_ab.py $p 64 2>&1 | grep -v amdgpu.ids
  python tools/step_ab.py $p 64 2>&1 | grep -v amdgpu.ids
done
echo "== loops, alternating (A = round 4)"
bash tools/ab_loop.sh "f16x3 bf16 bf16x3" 64
echo "== B = 32"
bash tools/ab_loop.sh "f16x3 bf16" 32
} > gpurun_out/r05/ab_deferred_ln_c03.txt 2>&1
cat gpurun_out/r05/ab_deferred_ln_c03.txt
