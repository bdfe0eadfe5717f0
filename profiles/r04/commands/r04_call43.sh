#!/bin/bash
# f32: X waves start their MFMA streams 32 nq cycles apart (TAMF_CLIP_STAGGER = 4, default build here) against 0 (G0) and 64 nq (G8)
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
L=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib
{ for i in 1 2 3; do
    echo "--- stagger 4"; python tools/loop_time.py f32 64 200 3
    echo "--- G0";  TAMF_LIB_OVERRIDE=$L/libtamf_hip_G0.so python tools/loop_time.py f32 64 200 3
    echo "--- G8";  TAMF_LIB_OVERRIDE=$L/libtamf_hip_G8.so python tools/loop_time.py f32 64 200 3
  done
} 2>&1 | grep -v amdgpu > gpurun_out/r04/stagger_c43.txt
cat gpurun_out/r04/stagger_c43.txt
