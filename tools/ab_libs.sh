#!/bin/bash
# GPU: whole-loop ms per DDPM step (and the GEMM hook timings) of several builds lib/libtamf_hip_<tag>.so against the working-tree
# build, alternating:  tools/ab_libs.sh "X2 X3" [prec] [B]
L=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib
prec=${2:-f16x3}; B=${3:-64}
for rep in 1 2; do
  python tools/loop_time.py $prec $B 200 3 2>&1 | grep ms/step
  for t in $1; do TAMF_LIB_OVERRIDE=$L/libtamf_hip_$t.so python tools/loop_time.py $prec $B 200 3 2>&1 | grep ms/step; done
done
echo "default:"; python tools/kbench.py $prec -1 2>&1 | grep -v amdgpu.ids | grep "ffn2 \|ffn1"
for t in $1; do echo "$t:"; TAMF_LIB_OVERRIDE=$L/libtamf_hip_$t.so python tools/kbench.py $prec -1 2>&1 | grep -v amdgpu.ids | grep "ffn2 \|ffn1"; done
