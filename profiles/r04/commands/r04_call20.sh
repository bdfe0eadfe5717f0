#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
{ python tools/step_ab.py bf16 64 -1,0x100fffff,0x110fffff,-1,0x100fffff,0x110fffff,-1 196; for t in -1 0x100fffff -1 0x100fffff; do python tools/loop_time.py bf16 64 200 3 $t 196; done; } 2>&1 | grep -E "variant|ms/step" > gpurun_out/r04/bf16_b64_ffn2_forms_c20.txt
cat gpurun_out/r04/bf16_b64_ffn2_forms_c20.txt
