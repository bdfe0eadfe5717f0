#!/bin/bash
# Overlap probe: an UPPER BOUND of what removing the kernel boundaries of a step could gain.  -DTAMF_BENCH build,
# plain launches; selection bit 256 lets the launches of the loop alternate between two streams with no data dependency enforced
# (garbage samples, time only).  Baseline = the same build, same plain launches, one stream.
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
L=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib/libtamf_hip_bench.so
{
for rep in 1 2; do
for p in f16x3 bf16 f32; do
  TAMF_LIB_OVERRIDE=$L timeout 300 python tools/loop_time.py $p 64 200 3 -1 196 nograph 2>&1 | grep ms/step
  TAMF_LIB_OVERRIDE=$L timeout 300 python tools/loop_time.py $p 64 200 3 0x100fffff 196 nograph 2>&1 | grep ms/step
done
done
for p in f16x3 bf16; do
  TAMF_LIB_OVERRIDE=$L timeout 300 python tools/loop_time.py $p 32 200 3 -1 196 nograph 2>&1 | grep ms/step
  TAMF_LIB_OVERRIDE=$L timeout 300 python tools/loop_time.py $p 32 200 3 0x100fffff 196 nograph 2>&1 | grep ms/step
  timeout 300 python tools/loop_time.py $p 64 200 3 -1 196 2>&1 | grep ms/step
done
} > gpurun_out/r06/overlap_probe_c02.txt 2>&1
cat gpurun_out/r06/overlap_probe_c02.txt
