#!/bin/bash
# LDS-DMA requests with scalar bases (no per-piece VALU) + f32 X-wave priorities: product build against P0 (same, priorities off) and OLD (HEAD)
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
L=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib
{ timeout 600 python -m pytest tests/test_hip_fullsize.py -x -q -m gpu -k "selections or exact_integers or b64_t196_vs_oracle or b64_t160_vs_oracle" 2>&1 | tail -3
  echo "=== f32 FFN1 timeline"; TAMF_LIB_OVERRIDE=$L/libtamf_hip_TL.so python tools/clip_timeline.py f32 0 2048 512 -1
  echo "=== bf16 FFN1 timeline"; TAMF_LIB_OVERRIDE=$L/libtamf_hip_TL.so python tools/clip_timeline.py bf16 0 2048 512 -1
  echo "=== bf16 FFN2 timeline"; TAMF_LIB_OVERRIDE=$L/libtamf_hip_TL.so python tools/clip_timeline.py bf16 3 512 2048 -1
  echo "=== f16x3 FFN1 timeline"; TAMF_LIB_OVERRIDE=$L/libtamf_hip_TL.so python tools/clip_timeline.py f16x3 0 2048 512 -1
  for prec in f32 bf16 f16x3 bf16x3; do for i in 1 2; do
    echo "--- new"; python tools/loop_time.py $prec 64 200 3
    echo "--- P0";  TAMF_LIB_OVERRIDE=$L/libtamf_hip_P0.so python tools/loop_time.py $prec 64 200 3
    echo "--- OLD";  TAMF_LIB_OVERRIDE=$L/libtamf_hip_OLD.so python tools/loop_time.py $prec 64 200 3
  done; done
  for prec in f32 bf16 f16x3; do
    echo "--- new B=32"; python tools/loop_time.py $prec 32 200 3
    echo "--- OLD B=32";  TAMF_LIB_OVERRIDE=$L/libtamf_hip_OLD.so python tools/loop_time.py $prec 32 200 3
    echo "--- new T=160"; python tools/loop_time.py $prec 64 200 3 -1 160
    echo "--- OLD T=160";  TAMF_LIB_OVERRIDE=$L/libtamf_hip_OLD.so python tools/loop_time.py $prec 64 200 3 -1 160
  done
} 2>&1 | grep -v amdgpu > gpurun_out/r04/dma_saddr_c32.txt
cat gpurun_out/r04/dma_saddr_c32.txt
