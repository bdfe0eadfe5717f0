#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
L=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib
{
for v in 0 3 4 5; do
  lib=$L/libtamf_hip_RB$v.so; [ $v = 0 ] && lib=$L/libtamf_hip.so
  for p in f16x3 bf16; do
    echo -n "ABL=$v rowblock "; TAMF_LIB_OVERRIDE=$lib python tools/kbench_one.py $p 2 0x7ffff 13312 512 2048 20 2>&1 | tail -1
    echo -n "ABL=$v rowblock "; TAMF_LIB_OVERRIDE=$lib python tools/kbench_one.py $p 2 0x7ffff 13312 512 512 20 2>&1 | tail -1
  done
done
for p in f16x3 bf16; do echo -n "default LN tile "; python tools/kbench_one.py $p 2 -1 13312 512 2048 20 2>&1 | tail -1; done
} > gpurun_out/r04/rowblock_ablation_c10.txt 2>&1
cat gpurun_out/r04/rowblock_ablation_c10.txt
python -m pytest tests/test_hip_fullsize.py -m gpu -x -q -k "rowblock" > gpurun_out/r04/gpu_tests_c10.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/gpu_tests_c10.log
grep -E "passed|failed|rc=|Error|assert" gpurun_out/r04/gpu_tests_c10.log | tail -5
