#!/bin/bash
# the whole GPU suite with per-test durations (735 s of the driver's 1 200 s step - where do they go?)
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
( time timeout 1500 python -m pytest tests/ -q -m gpu --durations=80 ) > gpurun_out/r06/gpu_tests_full_durations_c04.log 2>&1
tail -100 gpurun_out/r06/gpu_tests_full_durations_c04.log
