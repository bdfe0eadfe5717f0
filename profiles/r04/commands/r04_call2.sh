#!/bin/bash
# round 4, call 2: T = 160 clip tiles (NSUB = 11, parts of 6) + NKB = 6 attention: new parity tests, loop times, per-kernel step profiles
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
python -m pytest tests/test_hip_fullsize.py -m gpu -x -q -k "t160 or exact_integers or selections or clip_in_b64" > gpurun_out/r04/gpu_tests_c2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r04/gpu_tests_c2.log
grep -E "passed|failed|rc=|Error|assert" gpurun_out/r04/gpu_tests_c2.log | tail -8
{
for p in f16x3 bf16 f32; do python tools/loop_time.py $p 64 100 2 -1 196; python tools/loop_time.py $p 64 100 2 -1 160; done
for p in f16x3 bf16; do python tools/loop_time.py $p 32 100 2 -1 196; python tools/loop_time.py $p 32 100 2 -1 160; done
} 2>&1 | grep ms/step > gpurun_out/r04/loop_times_c2.txt
cat gpurun_out/r04/loop_times_c2.txt
for dt in f16x3 bf16 f32; do for T in 160 196; do
  python bench.py --steps 1 --warmup 0 --ddpm-steps 50 --no-cpu-baseline --also "" --fp32-loops 0 --check-clips 0 --dtype $dt --frames $T --profile-out gpurun_out/r04/step_profile_${dt}_T$T.json > /dev/null 2>&1
done; done
for B in 32; do for dt in f16x3 bf16; do
  python bench.py --steps 1 --warmup 0 --ddpm-steps 50 --no-cpu-baseline --also "" --fp32-loops 0 --check-clips 0 --dtype $dt --batch $B --profile-out gpurun_out/r04/step_profile_${dt}_B${B}_T196.json > /dev/null 2>&1
done; done
ls gpurun_out/r04
