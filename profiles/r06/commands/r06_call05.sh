#!/bin/bash
# (1) the GPU suite after sharing contexts / oracle loops across parametrisations (target: <= 600 s on the driver's box)
# (2) re-fetch A/B: the FFN hidden activations stored non-temporally (-DTAMF_H_NT build) against the default, whole loop + FFN1 / FFN2 traffic
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
( time timeout 1500 python -m pytest tests/ -q -m gpu --durations=25 ) > gpurun_out/r06/gpu_tests_full_c05.log 2>&1
tail -45 gpurun_out/r06/gpu_tests_full_c05.log
L=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib/libtamf_hip_hnt.so
{
for rep in 1 2; do
for p in f16x3 bf16 bf16x3; do
  TAMF_LIB_OVERRIDE=$L timeout 300 python tools/loop_time.py $p 64 200 3 2>&1 | grep ms/step
  timeout 300 python tools/loop_time.py $p 64 200 3 2>&1 | grep ms/step
done
done
for p in f16x3 bf16; do
  TAMF_LIB_OVERRIDE=$L timeout 300 python tools/loop_time.py $p 32 200 3 2>&1 | grep ms/step
  timeout 300 python tools/loop_time.py $p 32 200 3 2>&1 | grep ms/step
done
for p in f16x3 bf16; do
  for k in EpiBiasAct EpiResid EpiQKV; do
    echo "== $p $k  nt build"
    TAMF_LIB_OVERRIDE=$L bash tools/pmc_generic.sh $k "FETCH_SIZE|WRITE_SIZE" -- python3 tools/loop_time.py $p 64 10 1
    echo "== $p $k  default build"
    bash tools/pmc_generic.sh $k "FETCH_SIZE|WRITE_SIZE" -- python3 tools/loop_time.py $p 64 10 1
  done
done
} > gpurun_out/r06/ab_h_nt_c05.txt 2>&1
cat gpurun_out/r06/ab_h_nt_c05.txt
