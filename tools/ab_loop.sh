#!/bin/bash
# GPU: whole-loop ms per DDPM step of the A build (lib/libtamf_hip_A.so, tools/ab_build.sh) against the working tree, alternating
A=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib/libtamf_hip_A.so
for rep in 1 2; do
  for p in ${1:-f16x3 bf16 f32}; do
    TAMF_LIB_OVERRIDE=$A python tools/loop_time.py $p ${2:-64} 200 3 2>&1 | grep ms/step
    python tools/loop_time.py $p ${2:-64} 200 3 2>&1 | grep ms/step
  done
done
