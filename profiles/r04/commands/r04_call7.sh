#!/bin/bash
# rowblock kernel ablations (libs built with -DTAMF_RB_ABL=n: tools/ab_build.sh WORKTREE RBn) + the energy-model measurements
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
L=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib
{
for v in 0 1 2 3 4; do
  lib=$L/libtamf_hip_RB$v.so; [ $v = 0 ] && lib=$L/libtamf_hip.so
  for p in f16x3 bf16; do
    echo -n "ABL=$v "; TAMF_LIB_OVERRIDE=$lib python tools/kbench_one.py $p 2 -1 13312 512 2048 20 2>&1 | tail -1
    echo -n "ABL=$v "; TAMF_LIB_OVERRIDE=$lib python tools/kbench_one.py $p 2 -1 13312 512 512 20 2>&1 | tail -1
  done
done
echo "old LN tile (tuning 0x7ffff):"
for p in f16x3 bf16; do python tools/kbench_one.py $p 2 0x7ffff 13312 512 2048 20 | tail -1; python tools/kbench_one.py $p 2 0x7ffff 13312 512 512 20 | tail -1; done
} > gpurun_out/r04/rowblock_ablation_c7.txt 2>&1
cat gpurun_out/r04/rowblock_ablation_c7.txt
bash tools/energy_model.sh 5 > gpurun_out/r04/energy_run.log 2>&1
tail -30 gpurun_out/r04/energy_run.log
