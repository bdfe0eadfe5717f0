#!/bin/bash
# X waves of the clip GEMMs at priority 3 outside their MFMA stream (TAMF_CLIP_XPRIO, default 1) against the round-3 form (P0 build): timeline + whole loop, alternating
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
L=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib
{ echo "=== f32 FFN1 timeline, X priority flips"; TAMF_LIB_OVERRIDE=$L/libtamf_hip_TL.so python tools/clip_timeline.py f32 0 2048 512 -1
  echo "=== bf16 FFN1 timeline, X priority flips"; TAMF_LIB_OVERRIDE=$L/libtamf_hip_TL.so python tools/clip_timeline.py bf16 0 2048 512 -1
  echo "=== f16x3 FFN2 timeline, X priority flips"; TAMF_LIB_OVERRIDE=$L/libtamf_hip_TL.so python tools/clip_timeline.py f16x3 3 512 2048 -1
  for prec in f32 bf16 f16x3; do for i in 1 2; do
    echo "--- new"; python tools/loop_time.py $prec 64 200 3
    echo "--- P0";  TAMF_LIB_OVERRIDE=$L/libtamf_hip_P0.so python tools/loop_time.py $prec 64 200 3
  done; done
  for prec in f32 bf16 f16x3; do
    echo "--- new B=32"; python tools/loop_time.py $prec 32 200 3
    echo "--- P0 B=32";  TAMF_LIB_OVERRIDE=$L/libtamf_hip_P0.so python tools/loop_time.py $prec 32 200 3
  done
} 2>&1 | grep -v amdgpu > gpurun_out/r04/xprio_c30.txt
cat gpurun_out/r04/xprio_c30.txt
