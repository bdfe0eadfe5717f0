#!/bin/bash
# round profiles at 2853129: rocprofv3 kernel stats, FETCH/WRITE traffic, HIP-event step profiles, bench lines (default, T=160, config 3/5 shards), PMC totals
export TMPDIR=/tmp
export TAMF_ROUND=r06 TAMF_COMMIT=2853129
bash tools/round_profiles_all.sh > gpurun_out/round_profiles_all_c06.log 2>&1
tail -30 gpurun_out/round_profiles_all_c06.log
timeout 600 python3 bench.py --config 4 --no-cpu-baseline > gpurun_out/prof/bench_config4.log 2>&1; tail -n 1 gpurun_out/prof/bench_config4.log | cut -c1-300
timeout 600 python3 bench.py --batch 1 --frames 160 --also f32 --fp32-loops 1 --no-cpu-baseline --no-torch-baseline --steps 2 --warmup 1 > gpurun_out/prof/bench_b1_t160.log 2>&1; tail -n 1 gpurun_out/prof/bench_b1_t160.log | cut -c1-300
