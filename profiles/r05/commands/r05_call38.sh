#!/bin/bash
# round-5 final artefacts at 1e21e24 (every mode on the deferred LayerNorm; conditioning precompute in two launches): the full GPU suite, all profiles (tools/round_profiles_all.sh), bench --config 4 /
# B = 1, the f32 shape sweep again (its cliffs outside the clip-tile families came from the
# LayerNorm-fused tile), stress-fixture and 1000-step-loop error numbers
export TMPDIR=/tmp
export TAMF_COMMIT=1e21e24 TAMF_ROUND=r05
mkdir -p gpurun_out/r05 gpurun_out/prof
( time timeout 3000 python -m pytest tests -q -m gpu ) > gpurun_out/r05/gpu_tests_full_c38.log 2>&1
tail -4 gpurun_out/r05/gpu_tests_full_c38.log
bash tools/round_profiles_all.sh > gpurun_out/r05/round_profiles_all.log 2>&1
timeout 600 python bench.py --config 4 > gpurun_out/prof/bench_config4.json 2> gpurun_out/prof/bench_config4.err
timeout 600 python bench.py --batch 1 --frames 160 --also f32 > gpurun_out/prof/bench_b1_t160.json 2> gpurun_out/prof/bench_b1_t160.err
python tools/shape_sweep.py f32 arch_mdm_l 2>&1 | grep -v amdgpu.ids > gpurun_out/r05/shape_sweep_f32_c38.txt
timeout 1500 python -m pytest tests/test_hip_forward.py -q -s -m gpu -k "stress or loop_1000" 2>&1 | grep -E "stress|1000-step|passed|failed" > gpurun_out/r05/stress_and_loop_errors_c38.txt
tail -n 1 gpurun_out/prof/bench_default.log | cut -c1-600
tail -5 gpurun_out/prof_collect.log
cat gpurun_out/r05/shape_sweep_f32_c38.txt | cut -c1-300
