#!/bin/bash
# f32: per-wave phase stamps of the clip GEMMs' K-tile intervals (where do the ~20 % between the K loop and the MFMA peak go?)
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
L=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib
{ echo "=== f32 FFN1 (256-column tiles, K = 512)"; TAMF_LIB_OVERRIDE=$L/libtamf_hip_TL.so python tools/clip_timeline.py f32 0 2048 512 -1
  echo "=== f32 FFN2 (128-column tiles, K = 2048)"; TAMF_LIB_OVERRIDE=$L/libtamf_hip_TL.so python tools/clip_timeline.py f32 3 512 2048 -1
  echo "=== bf16 FFN1"; TAMF_LIB_OVERRIDE=$L/libtamf_hip_TL.so python tools/clip_timeline.py bf16 0 2048 512 -1
} 2>&1 | grep -v amdgpu > gpurun_out/r04/clip_timeline_f32_c29.txt
cat gpurun_out/r04/clip_timeline_f32_c29.txt
