#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
bash tools/energy_model.sh 5 > gpurun_out/r04/energy_run.log 2>&1
grep CASE gpurun_out/r04/energy_cases.txt | cut -c1-100 | head -8
head -3 gpurun_out/r04/energy_power_samples.txt | cut -c1-300
timeout 900 python bench.py > gpurun_out/r04/bench_default_c23.log 2>&1
tail -n 1 gpurun_out/r04/bench_default_c23.log | python -c "
import json,sys
l=json.loads(sys.stdin.read())
print({k:l[k] for k in ('value','ms_per_ddpm_step','check_ok')}, l.get('power'))
r=l['roofline']; print({k:r.get(k) for k in ('bound','frac','traffic_stale')}, (r.get('power_model') or {}).get('predicted_ms'))"
