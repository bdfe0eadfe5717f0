#!/bin/bash
# round 5, first contact: the new tests (sample.sh -> sample_refine.sh pipeline on the reference's file formats, guard-band sweep,
# module / forward suites), then the two bench lines the round-4 verdict asked for (config 4 = R trunk; B = 1, T = 160)
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
timeout 900 python -m pytest tests/test_hip_pipeline.py tests/test_hip_module.py -x -q -m gpu > gpurun_out/r05/gpu_tests_pipeline_c01.log 2>&1
tail -5 gpurun_out/r05/gpu_tests_pipeline_c01.log
( time timeout 1500 python -m pytest tests/test_hip_guardbands.py -x -q -m gpu ) > gpurun_out/r05/gpu_tests_guardbands_c01.log 2>&1
tail -8 gpurun_out/r05/gpu_tests_guardbands_c01.log
timeout 600 python bench.py --config 4 > gpurun_out/r05/bench_config4_c01.json 2> gpurun_out/r05/bench_config4_c01.err
cut -c1-1200 gpurun_out/r05/bench_config4_c01.json; tail -3 gpurun_out/r05/bench_config4_c01.err
timeout 600 python bench.py --batch 1 --frames 160 --also f32 > gpurun_out/r05/bench_b1_t160_c01.json 2> gpurun_out/r05/bench_b1_t160_c01.err
cut -c1-600 gpurun_out/r05/bench_b1_t160_c01.json; tail -3 gpurun_out/r05/bench_b1_t160_c01.err
