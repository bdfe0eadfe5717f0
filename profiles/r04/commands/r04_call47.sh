#!/bin/bash
# final artefacts of the round at db0cafb: all profiles (tools/round_profiles_all.sh), the issue_overlap microbenchmark, the full GPU suite
export TMPDIR=/tmp
export TAMF_COMMIT=db0cafb
mkdir -p gpurun_out/r04

bash tools/round_profiles_all.sh > gpurun_out/r04/round_profiles_all.log 2>&1
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r04/gpu_tests_final_db0cafb.log 2>&1
tail -3 gpurun_out/r04/gpu_tests_final_db0cafb.log
tail -n 1 gpurun_out/prof/bench_default.log | cut -c1-1500
