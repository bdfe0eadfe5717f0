#!/bin/bash
# deferred LayerNorm with the CENTRING folded into the weights (W'' = W diag(gamma) (I - 1 1^T / d): one fma per element in the consumers'
# epilogues): parity (forward / robustness / fullsize subsets), standalone GEMM timings old vs new forms, loops against round 4 (lib A)
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
A=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib/libtamf_hip_A.so
{
python __graft_entry__.py smoke 2>&1 | tail -4
timeout 900 python -m pytest tests/test_hip_forward.py tests/test_hip_robustness.py -x -q -m gpu 2>&1 | tail -4
timeout 1200 python -m pytest tests/test_hip_fullsize.py -x -q -m gpu -k "equals_clip_alone or vs_oracle or same_bits" 2>&1 | tail -3
python tools/kbench.py f16x3,bf16 -1 2>&1 | grep -v amdgpu.ids
for p in bf16 f16x3; do
  echo "== per-kernel, $p, B=64: working tree"
  python tools/step_ab.py $p 64 2>&1 | grep -v amdgpu.ids
done
echo "== loops, alternating (A = round 4)"
bash tools/ab_loop.sh "f16x3 bf16 bf16x3" 64
bash tools/ab_loop.sh "f16x3 bf16" 32
python tests/scripts/parity_report.py 2>&1 | tail -30
} > gpurun_out/r05/ab_deferred_ln_c09.txt 2>&1
cat gpurun_out/r05/ab_deferred_ln_c09.txt | cut -c1-250
