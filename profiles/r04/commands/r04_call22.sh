#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
A=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib/libtamf_hip_A.so
{ echo "A = working tree built with -DTAMF_NO_PRESCALE (unscaled f16x3 weights: subnormal lo planes); default = pre-scaled weights"
for rep in 1 2 3; do
  TAMF_LIB_OVERRIDE=$A python tools/loop_time.py f16x3 64 200 3 -1 196 2>&1 | grep ms/step
  python tools/loop_time.py f16x3 64 200 3 -1 196 2>&1 | grep ms/step
done; } > gpurun_out/r04/ab_prescale_energy_c22.txt 2>&1
cat gpurun_out/r04/ab_prescale_energy_c22.txt
