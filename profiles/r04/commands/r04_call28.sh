#!/bin/bash
# f32: the in_proj GEMM as ONE clip launch (Q | K and V tiles mixed in a workgroup's stream) against the two launches (selection 0x80)
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
{ timeout 900 python -m pytest tests/test_hip_fullsize.py tests/test_hip_forward.py -x -q -m gpu -k "f32" 2>&1 | tail -5
  python tools/step_ab.py f32 64 -1,0x080fffff 196
  python tools/step_ab.py f32 64 -1,0x080fffff 160
  for i in 1 2; do python tools/loop_time.py f32 64 200 3 -1 196; python tools/loop_time.py f32 64 200 3 0x080fffff 196; done
  python tools/loop_time.py f32 64 200 3 -1 160; python tools/loop_time.py f32 64 200 3 0x080fffff 160
} > gpurun_out/r04/mixed_qkv_c28.txt 2>&1
cat gpurun_out/r04/mixed_qkv_c28.txt
