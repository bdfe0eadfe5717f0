#!/bin/bash
# Y waves of the 128-column clip GEMMs read the next K tile's fragments inside their MFMA stream (TAMF_CLIP_YFUSE, default) against the separate read phase (Y0)
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
L=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib
{ timeout 900 python -m pytest tests/test_hip_fullsize.py -x -q -m gpu -k "selections or exact_integers or b64_t196_vs_oracle or b64_t160_vs_oracle or other_batch" 2>&1 | tail -3
  echo "=== f32 FFN2 timeline"; TAMF_LIB_OVERRIDE=$L/libtamf_hip_TL.so python tools/clip_timeline.py f32 3 512 2048 -1
  echo "=== f16x3 FFN2 timeline"; TAMF_LIB_OVERRIDE=$L/libtamf_hip_TL.so python tools/clip_timeline.py f16x3 3 512 2048 -1
  for prec in f32 f16x3 bf16 bf16x3; do for i in 1 2; do
    echo "--- new"; python tools/loop_time.py $prec 64 200 3
    echo "--- Y0";  TAMF_LIB_OVERRIDE=$L/libtamf_hip_Y0.so python tools/loop_time.py $prec 64 200 3
  done; done
  for prec in f32 f16x3; do
    echo "--- new B=32"; python tools/loop_time.py $prec 32 200 3
    echo "--- Y0 B=32";  TAMF_LIB_OVERRIDE=$L/libtamf_hip_Y0.so python tools/loop_time.py $prec 32 200 3
    echo "--- new T=160"; python tools/loop_time.py $prec 64 200 3 -1 160
    echo "--- Y0 T=160";  TAMF_LIB_OVERRIDE=$L/libtamf_hip_Y0.so python tools/loop_time.py $prec 64 200 3 -1 160
  done
} 2>&1 | grep -v amdgpu > gpurun_out/r04/yfuse_c36.txt
cat gpurun_out/r04/yfuse_c36.txt
