#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
bash tools/energy_model.sh 5 > gpurun_out/r04/energy_run.log 2>&1
grep CASE gpurun_out/r04/energy_cases.txt | cut -c1-100 | tail -6
python -m pytest tests -m gpu -x -q > gpurun_out/r04/gpu_tests_c14.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/gpu_tests_c14.log
grep -E "passed|failed|rc=|Error|assert" gpurun_out/r04/gpu_tests_c14.log | tail -5
