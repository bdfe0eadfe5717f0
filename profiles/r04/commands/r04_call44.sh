#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
timeout 1500 python -m pytest tests/test_hip_fullsize.py -x -q -m gpu -s -k "largest_padding" 2>&1 | grep -E "max-padding|passed|failed|Error" > gpurun_out/r04/max_padding_tests_c44.txt
cat gpurun_out/r04/max_padding_tests_c44.txt
