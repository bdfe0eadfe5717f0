#!/bin/bash
# the new GPU test: calls of a few / 20 - 39 clips give the bits of the big-batch tiles
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
timeout 1200 python -m pytest tests/test_hip_fullsize.py -q -m gpu -k "small_and_mid" 2>&1 | tail -5 > gpurun_out/r05/gpu_test_small_and_mid_c40.log
cat gpurun_out/r05/gpu_test_small_and_mid_c40.log
