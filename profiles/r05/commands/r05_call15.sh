#!/bin/bash
# f32 with the LayerNorms deferred too (EpiQK / EpiVt with the row factors, EpiResid for f32): parity subsets, per-kernel profile,
# loops against the c4cd61f library (lib A: f32 = GEMM + LayerNorm kernel) alternating on one box
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
{
python __graft_entry__.py smoke 2>&1 | tail -4
timeout 900 python -m pytest tests/test_hip_forward.py tests/test_hip_robustness.py -x -q -m gpu 2>&1 | tail -4
timeout 1200 python -m pytest tests/test_hip_fullsize.py -x -q -m gpu -k "equals_clip_alone or vs_oracle or same_bits" 2>&1 | tail -3
echo "== per-kernel, f32, B=64: working tree"
python tools/step_ab.py f32 64 2>&1 | grep -v amdgpu.ids
echo "== loops, alternating (A = c4cd61f)"
bash tools/ab_loop.sh "f32" 64
bash tools/ab_loop.sh "f32" 32
python tests/scripts/parity_report.py 2>&1 | grep -i "f32\|mode" | tail -30
} > gpurun_out/r05/ab_f32_deferred_c15.txt 2>&1
tail -40 gpurun_out/r05/ab_f32_deferred_c15.txt
