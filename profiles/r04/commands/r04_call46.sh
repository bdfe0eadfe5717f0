#!/bin/bash
# f32: GELU through the Abramowitz-Stegun erf (what the other modes use) instead of libm's erff: FFN1 launch, whole loop, error vs the default build
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
L=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib
{ python tools/step_ab.py f32 64 -1 196
  TAMF_LIB_OVERRIDE=$L/libtamf_hip_GF.so python tools/step_ab.py f32 64 -1 196
  for i in 1 2; do
    echo "--- erff"; python tools/loop_time.py f32 64 200 3
    echo "--- erf_as";  TAMF_LIB_OVERRIDE=$L/libtamf_hip_GF.so python tools/loop_time.py f32 64 200 3
  done
} 2>&1 | grep -v amdgpu > gpurun_out/r04/f32_gelu_c46.txt
cat gpurun_out/r04/f32_gelu_c46.txt
