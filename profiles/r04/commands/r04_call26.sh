#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
L=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib
{ echo "A = -DTAMF_MMA_INTERLEAVE=0 (Op::mma chains everywhere); default = f32 interleaved; B = =15 (every mode interleaved)"
for rep in 1 2; do
  for p in f32 f16x3 bf16; do
    TAMF_LIB_OVERRIDE=$L/libtamf_hip_A.so python tools/loop_time.py $p 64 100 2 -1 196 2>&1 | grep ms/step
    python tools/loop_time.py $p 64 100 2 -1 196 2>&1 | grep ms/step
    TAMF_LIB_OVERRIDE=$L/libtamf_hip_B.so python tools/loop_time.py $p 64 100 2 -1 196 2>&1 | grep ms/step
  done
done
python tools/step_ab.py f32 64 -1 196; TAMF_LIB_OVERRIDE=$L/libtamf_hip_A.so python tools/step_ab.py f32 64 -1 196
} > gpurun_out/r04/ab_mma_interleave_c26.txt 2>&1
grep -v amdgpu.ids gpurun_out/r04/ab_mma_interleave_c26.txt
