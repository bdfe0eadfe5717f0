#!/bin/bash
# attention key split with the four light waves placed on the SIMDs that carry the fewest tiles of the last workgroup: tests, then A/B against K0 (no key split)
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
L=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib
{ timeout 1500 python -m pytest tests/test_hip_fullsize.py tests/test_hip_kernels.py -x -q -m gpu 2>&1 | tail -3
  for prec in f32 f16x3 bf16; do for B in 64 32 16; do
    echo "--- new $prec B=$B"; python tools/loop_time.py $prec $B 200 3
    echo "--- K0 $prec B=$B";  TAMF_LIB_OVERRIDE=$L/libtamf_hip_K0.so python tools/loop_time.py $prec $B 200 3
  done; done
} 2>&1 | grep -v amdgpu > gpurun_out/r04/attn_ksplit_c48.txt
cat gpurun_out/r04/attn_ksplit_c48.txt
