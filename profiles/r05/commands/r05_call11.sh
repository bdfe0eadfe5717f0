#!/bin/bash
# the numbers of DESIGN.md section 2 (stress fixtures incl. the new DC-offset one, 1000-step loops: pytest -s prints them), and short clips
# (T = 64: the attention instantiations that spilled 880 - 1 048 bytes per lane in round 4) against the round-4 library
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
A=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib/libtamf_hip_A.so
timeout 1500 python -m pytest tests/test_hip_forward.py -q -s -m gpu -k "stress or loop_1000" 2>&1 | grep -E "stress|1000-step|passed|failed" > gpurun_out/r05/stress_and_loop_errors_c11.txt
cat gpurun_out/r05/stress_and_loop_errors_c11.txt | cut -c1-220
{
echo "== T = 64, B = 64 (arch_mdm_l): loops, A = round 4"
for rep in 1 2; do for p in f16x3 bf16 f32; do
  TAMF_LIB_OVERRIDE=$A python tools/loop_time.py $p 64 200 3 -1 64 2>&1 | grep ms/step
  python tools/loop_time.py $p 64 200 3 -1 64 2>&1 | grep ms/step
done; done
echo "== per-kernel T = 64"
for p in f16x3 bf16; do
TAMF_LIB_OVERRIDE=$A python tools/step_ab.py $p 64 -1 64 2>&1 | grep -v amdgpu.ids
python tools/step_ab.py $p 64 -1 64 2>&1 | grep -v amdgpu.ids
done
} > gpurun_out/r05/ab_short_clips_t64_c11.txt 2>&1
cat gpurun_out/r05/ab_short_clips_t64_c11.txt | cut -c1-250
