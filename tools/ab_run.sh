#!/bin/bash
# GPU: alternate the A build (lib/libtamf_hip_A.so, see ab_build.sh) and the working-tree build on the same box
A=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib/libtamf_hip_A.so
for rep in 1 2; do
  echo "--- A (reference build)"; TAMF_LIB_OVERRIDE=$A python tools/kbench.py ${1:-bf16x3,bf16} -1 2>&1 | grep -v amdgpu
  echo "--- B (working tree)";    python tools/kbench.py ${1:-bf16x3,bf16} -1 2>&1 | grep -v amdgpu
done
