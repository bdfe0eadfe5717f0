#!/bin/bash
# final artefacts of the round at 3dd4e67: all profiles (tools/round_profiles_all.sh), the issue_overlap microbenchmark, the full GPU suite
export TMPDIR=/tmp
export TAMF_COMMIT=3dd4e67
mkdir -p gpurun_out/r04

bash tools/round_profiles_all.sh > gpurun_out/r04/round_profiles_all.log 2>&1
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r04/gpu_tests_final_3dd4e67.log 2>&1
tail -3 gpurun_out/r04/gpu_tests_final_3dd4e67.log
tail -n 1 gpurun_out/prof/bench_default.log | cut -c1-1500
