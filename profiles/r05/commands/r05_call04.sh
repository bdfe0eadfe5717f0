#!/bin/bash
# deferred LayerNorm, second pass (row terms staged as (ra, rb), two tiles staged up front, residual prefetch): parity of the fullsize
# suite's invariance tests + same-box A/B against round 4 (lib A)
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
A=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib/libtamf_hip_A.so
timeout 1200 python -m pytest tests/test_hip_fullsize.py -x -q -m gpu -k "equals_clip_alone or same_bits or vs_oracle or over_batch" > gpurun_out/r05/gpu_tests_fullsize_c04.log 2>&1; tail -4 gpurun_out/r05/gpu_tests_fullsize_c04.log
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
} > gpurun_out/r05/ab_deferred_ln_c04.txt 2>&1
cat gpurun_out/r05/ab_deferred_ln_c04.txt | cut -c1-250
