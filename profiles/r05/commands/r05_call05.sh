#!/bin/bash
# deferred LayerNorm, third pass (statistics of all passes requested up front, first two tiles staged by all eight waves): standalone
# GEMM timings of the old and new forms in ONE process (tools/kbench.py), quick invariance tests, loops against round 4
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
A=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib/libtamf_hip_A.so
{
python tools/kbench.py f16x3,bf16 -1 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests/test_hip_fullsize.py -x -q -m gpu -k "equals_clip_alone or b64_t196_vs_oracle or b64_t160_vs_oracle" 2>&1 | tail -3
for p in bf16 f16x3; do
  echo "== per-kernel, $p, B=64: working tree"
  python tools/step_ab.py $p 64 2>&1 | grep -v amdgpu.ids
done
echo "== loops, alternating (A = round 4)"
bash tools/ab_loop.sh "f16x3 bf16" 64
} > gpurun_out/r05/ab_deferred_ln_c05.txt 2>&1
cat gpurun_out/r05/ab_deferred_ln_c05.txt | cut -c1-250
