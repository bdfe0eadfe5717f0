#!/bin/bash
# f32, 256-column clip tiles: X : Y = 7 : 6 row tiles with Y's fragment reads inside its MFMA stream (default) against 6 : 7 with the separate read phase (N47)
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
L=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib
{ timeout 900 python -m pytest tests/test_hip_fullsize.py -x -q -m gpu -k "f32 and (selections or exact_integers or b64_t196_vs_oracle or b64_t160_vs_oracle or other_batch)" 2>&1 | tail -3
  for i in 1 2 3; do
    echo "--- new"; python tools/loop_time.py f32 64 200 3
    echo "--- N47";  TAMF_LIB_OVERRIDE=$L/libtamf_hip_N47.so python tools/loop_time.py f32 64 200 3
  done
  echo "--- new B=32"; python tools/loop_time.py f32 32 200 3
  echo "--- N47 B=32";  TAMF_LIB_OVERRIDE=$L/libtamf_hip_N47.so python tools/loop_time.py f32 32 200 3
  echo "--- new T=160"; python tools/loop_time.py f32 64 200 3 -1 160
  echo "--- N47 T=160";  TAMF_LIB_OVERRIDE=$L/libtamf_hip_N47.so python tools/loop_time.py f32 64 200 3 -1 160
  python tools/step_ab.py f32 64 -1 196
  TAMF_LIB_OVERRIDE=$L/libtamf_hip_N47.so python tools/step_ab.py f32 64 -1 196
} 2>&1 | grep -v amdgpu > gpurun_out/r04/yfuse_n4_c37.txt
cat gpurun_out/r04/yfuse_n4_c37.txt
