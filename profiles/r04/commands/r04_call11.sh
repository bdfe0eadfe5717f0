#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
A=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib/libtamf_hip_A.so
{
echo "== QKV standalone (kbench): working tree (4 x 2 XCD arrangement in the persistent grid) vs A (round-robin order)"
for rep in 1 2; do for p in f16x3 bf16; do
  echo -n "tree "; python tools/kbench_one.py $p 1 -1 13312 1536 512 20 2>&1 | tail -1
  echo -n "A    "; TAMF_LIB_OVERRIDE=$A python tools/kbench_one.py $p 1 -1 13312 1536 512 20 2>&1 | tail -1
done; done
echo "== FETCH_SIZE / WRITE_SIZE (KiB; FETCH x2 per the gfx950 correction) of one QKV launch"
echo "tree:"; bash tools/pmc_generic.sh gemm_kernel "FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum TCC_MISS_sum" -- python3 tools/kbench_one.py f16x3 1 -1 13312 1536 512 5
echo "A:"; TAMF_LIB_OVERRIDE=$A bash tools/pmc_generic.sh gemm_kernel "FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum TCC_MISS_sum" -- python3 tools/kbench_one.py f16x3 1 -1 13312 1536 512 5
echo "== whole loop, alternating"
bash tools/ab_loop.sh "f16x3 bf16" 64
echo "== B = 32: QKV on clip tiles from 50 % utilisation (selection 0x480) vs default"
python tools/step_ab.py f16x3 32 -1,0x480fffff
python tools/step_ab.py bf16 32 -1,0x480fffff
} > gpurun_out/r04/qkv_xcd42_c11.txt 2>&1
grep -v "amdgpu.ids" gpurun_out/r04/qkv_xcd42_c11.txt
python -m pytest tests -m gpu -x -q > gpurun_out/r04/gpu_tests_c11.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/gpu_tests_c11.log
grep -E "passed|failed|rc=|Error|assert" gpurun_out/r04/gpu_tests_c11.log | tail -5
