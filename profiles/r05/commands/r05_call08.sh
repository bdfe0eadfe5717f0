#!/bin/bash
# round-5 artefacts at 9e6dc5a: all profiles (tools/round_profiles_all.sh: rocprofv3 kernel stats, PMC traffic, step profiles, bench lines,
# step totals, attention / GEMM counters, parity report), bench --config 4
export TMPDIR=/tmp
export TAMF_COMMIT=9e6dc5a TAMF_ROUND=r05
mkdir -p gpurun_out/r05
bash tools/round_profiles_all.sh > gpurun_out/r05/round_profiles_all.log 2>&1
timeout 600 python bench.py --config 4 > gpurun_out/prof/bench_config4.json 2> gpurun_out/prof/bench_config4.err
tail -n 1 gpurun_out/prof/bench_default.log | cut -c1-1500
ls gpurun_out/prof gpurun_out/prof/pmc | head -80
tail -5 gpurun_out/prof_collect.log
