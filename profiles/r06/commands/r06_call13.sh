#!/bin/bash
# end-of-round collection at 8a511b7 (the launch lock left the product build: the sources' digest moved once more): kernel stats, traffic, step profiles, bench lines,
# inter-kernel gaps of the hipGraph loop from a kernel trace, the whole GPU suite and smoke()
export TMPDIR=/tmp
export TAMF_ROUND=r06 TAMF_COMMIT=8a511b7
mkdir -p gpurun_out/r06
bash tools/round_profiles_all.sh > gpurun_out/round_profiles_all_c13.log 2>&1
tail -5 gpurun_out/round_profiles_all_c13.log
timeout 600 python3 bench.py --config 4 --no-cpu-baseline > gpurun_out/prof/bench_config4.log 2>&1; tail -n 1 gpurun_out/prof/bench_config4.log | cut -c1-200
timeout 600 python3 bench.py --batch 1 --frames 160 --also f32 --fp32-loops 1 --no-cpu-baseline --no-torch-baseline --steps 2 --warmup 1 > gpurun_out/prof/bench_b1_t160.log 2>&1; tail -n 1 gpurun_out/prof/bench_b1_t160.log | cut -c1-200
for p in f16x3 bf16; do
  rm -rf gpurun_out/kt
  timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/kt -o r -- python3 tools/loop_time.py $p 64 50 1 > gpurun_out/kt.log 2>&1
  f=$(find gpurun_out/kt -name "*kernel_trace.csv" | head -1)
  echo "== inter-kernel gaps, hipGraph loop, $p (rocprofv3 --kernel-trace; the first 300 dispatches - set-up and the first loop - skipped)" >> gpurun_out/r06/gap_report_c13.txt
  python3 tools/gap_report.py $f 2500 >> gpurun_out/r06/gap_report_c13.txt 2>&1
done
rm -rf gpurun_out/kt
cat gpurun_out/r06/gap_report_c13.txt
( time timeout 1500 python -m pytest tests/ -q -m gpu ) > gpurun_out/r06/gpu_tests_full_c13.log 2>&1
tail -12 gpurun_out/r06/gpu_tests_full_c13.log
timeout 300 python __graft_entry__.py smoke > gpurun_out/r06/smoke_c13.log 2>&1; tail -6 gpurun_out/r06/smoke_c13.log
