#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
{ for p in f16x3 bf16; do python tools/step_ab.py $p 64 -1,0x10fffff,0x100fffff,-1,0x10fffff 160; done; python tools/step_ab.py f16x3 32 -1,0x10fffff 160; } 2>&1 | grep variant > gpurun_out/r04/t160_selections_c17.txt
cat gpurun_out/r04/t160_selections_c17.txt
