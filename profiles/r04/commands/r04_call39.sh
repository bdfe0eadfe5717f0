#!/bin/bash
# attention: the last of a clip's 13 query tiles key-split over four waves, one per SIMD (TAMF_ATTN_KSPLIT, default) against the 13-wave form (K0)
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
L=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib
{ timeout 1500 python -m pytest tests/test_hip_fullsize.py tests/test_hip_forward.py tests/test_hip_kernels.py -x -q -m gpu 2>&1 | tail -5
  for prec in f32 f16x3 bf16; do
    python tools/step_ab.py $prec 64 -1 196
    TAMF_LIB_OVERRIDE=$L/libtamf_hip_K0.so python tools/step_ab.py $prec 64 -1 196
    for i in 1 2; do
      echo "--- new"; python tools/loop_time.py $prec 64 200 3
      echo "--- K0";  TAMF_LIB_OVERRIDE=$L/libtamf_hip_K0.so python tools/loop_time.py $prec 64 200 3
    done
    echo "--- new B=32"; python tools/loop_time.py $prec 32 200 3
    echo "--- K0 B=32";  TAMF_LIB_OVERRIDE=$L/libtamf_hip_K0.so python tools/loop_time.py $prec 32 200 3
  done
} 2>&1 | grep -v amdgpu > gpurun_out/r04/attn_ksplit_c39.txt
cat gpurun_out/r04/attn_ksplit_c39.txt
