#!/bin/bash
# f32: LDS-DMA requests between the X waves' own MFMAs (TAMF_CLIP_SPREAD = 2 per 8 MFMAs, default) against 1 per 8 (S1) and the batch (S0)
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
L=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib
{ timeout 600 python -m pytest tests/test_hip_fullsize.py -x -q -m gpu -k "f32 and (selections or exact_integers or b64_t196_vs_oracle or b64_t160_vs_oracle or other_batch)" 2>&1 | tail -3
  echo "=== f32 FFN1 timeline"; TAMF_LIB_OVERRIDE=$L/libtamf_hip_TL.so python tools/clip_timeline.py f32 0 2048 512 -1
  echo "=== f32 FFN2 timeline"; TAMF_LIB_OVERRIDE=$L/libtamf_hip_TL.so python tools/clip_timeline.py f32 3 512 2048 -1
  for i in 1 2; do
    echo "--- new (2 per group)"; python tools/loop_time.py f32 64 200 3
    echo "--- S1";  TAMF_LIB_OVERRIDE=$L/libtamf_hip_S1.so python tools/loop_time.py f32 64 200 3
    echo "--- S0";  TAMF_LIB_OVERRIDE=$L/libtamf_hip_S0.so python tools/loop_time.py f32 64 200 3
  done
  echo "--- new B=32"; python tools/loop_time.py f32 32 200 3
  echo "--- S0 B=32";  TAMF_LIB_OVERRIDE=$L/libtamf_hip_S0.so python tools/loop_time.py f32 32 200 3
  echo "--- new T=160"; python tools/loop_time.py f32 64 200 3 -1 160
  echo "--- S0 T=160";  TAMF_LIB_OVERRIDE=$L/libtamf_hip_S0.so python tools/loop_time.py f32 64 200 3 -1 160
  python tools/step_ab.py f32 64 -1 196
  TAMF_LIB_OVERRIDE=$L/libtamf_hip_S0.so python tools/step_ab.py f32 64 -1 196
} 2>&1 | grep -v amdgpu > gpurun_out/r04/dma_spread_c33.txt
cat gpurun_out/r04/dma_spread_c33.txt
