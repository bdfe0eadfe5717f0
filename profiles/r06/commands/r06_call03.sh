#!/bin/bash
# Overlap probe again with a build that carries ONLY the probe (-DTAMF_OVERLAP_PROBE; the -DTAMF_BENCH build of call 02 ran its bf16 / f32
# kernels 5 x slower - ablation code in the kernels), + the new GPU tests of the round so far
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
L=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib/libtamf_hip_probe.so
{
for rep in 1 2; do
for p in f16x3 bf16 f32 bf16x3; do
  TAMF_LIB_OVERRIDE=$L timeout 300 python tools/loop_time.py $p 64 200 3 -1 196 nograph 2>&1 | grep ms/step
  TAMF_LIB_OVERRIDE=$L timeout 300 python tools/loop_time.py $p 64 200 3 0x100fffff 196 nograph 2>&1 | grep ms/step
done
done
for p in f16x3 bf16; do
  TAMF_LIB_OVERRIDE=$L timeout 300 python tools/loop_time.py $p 32 200 3 -1 196 nograph 2>&1 | grep ms/step
  TAMF_LIB_OVERRIDE=$L timeout 300 python tools/loop_time.py $p 32 200 3 0x100fffff 196 nograph 2>&1 | grep ms/step
  timeout 300 python tools/loop_time.py $p 64 200 3 -1 196 2>&1 | grep ms/step
done
} > gpurun_out/r06/overlap_probe_c03.txt 2>&1
cat gpurun_out/r06/overlap_probe_c03.txt
timeout 900 python -m pytest tests/test_hip_guardbands.py::test_a_resize_that_runs_out_of_memory_leaves_the_context_working_at_its_old_size tests/test_hip_guardbands.py::test_the_guard_bands_do_catch_an_overrun tests/test_hip_module.py::test_module_text_branch_with_a_stub_clip tests/test_hip_forward.py -k "trained or resize or overrun or stub_clip" -x -q -s 2>&1 | tail -40 > gpurun_out/r06/gpu_tests_new_c03.log
cat gpurun_out/r06/gpu_tests_new_c03.log
