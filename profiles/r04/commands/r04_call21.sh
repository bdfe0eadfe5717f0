#!/bin/bash
# same-box A/B: round-3 library (29ea60d, lib/libtamf_hip_A.so) against the working tree, alternating
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
A=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib/libtamf_hip_A.so
{
for rep in 1 2; do
  for cfg in "f16x3 64 196" "f16x3 32 196" "f16x3 64 160" "f32 64 196" "f32 64 160" "f32 32 196" "bf16 64 196" "bf16 32 196" "bf16 64 160"; do
    set -- $cfg
    TAMF_LIB_OVERRIDE=$A python tools/loop_time.py $1 $2 100 2 -1 $3 2>&1 | grep ms/step
    python tools/loop_time.py $1 $2 100 2 -1 $3 2>&1 | grep ms/step
  done
done
} > gpurun_out/r04/ab_same_box_r03_29ea60d_vs_r04.log 2>&1
cat gpurun_out/r04/ab_same_box_r03_29ea60d_vs_r04.log
python __graft_entry__.py smoke 2>&1 | tail -4
