#!/bin/bash
# round-5 final artefacts at c4cd61f: the full GPU suite, then all profiles (tools/round_profiles_all.sh) and bench --config 4 / B = 1
export TMPDIR=/tmp
export TAMF_COMMIT=c4cd61f TAMF_ROUND=r05
mkdir -p gpurun_out/r05
( time timeout 3000 python -m pytest tests -q -m gpu ) > gpurun_out/r05/gpu_tests_full_c10.log 2>&1
tail -6 gpurun_out/r05/gpu_tests_full_c10.log
bash tools/round_profiles_all.sh > gpurun_out/r05/round_profiles_all.log 2>&1
timeout 600 python bench.py --config 4 > gpurun_out/prof/bench_config4.json 2> gpurun_out/prof/bench_config4.err
timeout 600 python bench.py --batch 1 --frames 160 --also f32 > gpurun_out/prof/bench_b1_t160.json 2> gpurun_out/prof/bench_b1_t160.err
tail -n 1 gpurun_out/prof/bench_default.log | cut -c1-400
tail -5 gpurun_out/prof_collect.log
