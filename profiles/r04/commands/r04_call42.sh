#!/bin/bash
# L2 touch-prefetch depth (tuning bits 8..11) on the generic 128 x 128 / 64 x 128 GEMMs of the step: input merges, QKV (16-bit modes), head
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
{ for prec in f16x3 bf16 f32; do python tools/step_ab.py $prec 64 -1,0x200,0x400,0x800 196; done
  python tools/step_ab.py f16x3 32 -1,0x200,0x400 196
} 2>&1 | grep -v amdgpu > gpurun_out/r04/touch_prefetch_c42.txt
cat gpurun_out/r04/touch_prefetch_c42.txt
