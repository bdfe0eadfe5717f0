#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
{ for p in f16x3 bf16 f32; do python tools/step_ab.py $p 32 -1,0x10fffff,-1,0x10fffff; done; } 2>&1 | grep variant > gpurun_out/r04/outproj_parts_b32_c15.txt
cat gpurun_out/r04/outproj_parts_b32_c15.txt
