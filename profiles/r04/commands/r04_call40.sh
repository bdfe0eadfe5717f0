#!/bin/bash
# f32 attention: what the kernel's time is made of (ablations of a -DTAMF_BENCH build: 1 no LDS-DMA, 2 no MFMAs, 4 no fragment reads, 8 no exp2, 16 no store)
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
L=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib
TAMF_LIB_OVERRIDE=$L/libtamf_hip_BN.so python tools/attn_bench.py f32,f16x3,bf16 0,1,2,4,8,16,6,7,14,22,31 -1 64 2>&1 | grep -v amdgpu > gpurun_out/r04/attn_ablation_f32_c40.txt
cat gpurun_out/r04/attn_ablation_f32_c40.txt
