#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
{ python tools/step_ab.py f32 64 -1,0x20fffff,0x400fffff; python tools/step_ab.py f32 64 -1,0x20fffff; } 2>&1 | grep variant > gpurun_out/r04/f32_ffn1_ab_c12.txt
cat gpurun_out/r04/f32_ffn1_ab_c12.txt
timeout 900 python bench.py > gpurun_out/r04/bench_default_c12.log 2>&1
tail -n 1 gpurun_out/r04/bench_default_c12.log | python -c "
import json,sys
l=json.loads(sys.stdin.read())
print({k:l[k] for k in ('value','ms_per_ddpm_step','dtype','check_ok')}, l.get('power'))
r=l['roofline']; print({k:r.get(k) for k in ('bound','kernel','achieved','frac','traffic','traffic_stale','bound_note')}); print(r.get('power_model')); print(r.get('reference_arithmetic'))
print(l['cpu_baseline'])"
