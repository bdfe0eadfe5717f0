#!/bin/bash
# config 4 (the R trunk, one forward per batch): where its 1.4 ms go, per launch
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
timeout 600 python bench.py --config 4 --no-cpu-baseline --also f32,bf16 > gpurun_out/r05/bench_config4_c20.json 2> gpurun_out/r05/bench_config4_c20.err
python3 - <<'PY'
import json
d=json.loads(open('gpurun_out/r05/bench_config4_c20.json').read().strip().splitlines()[-1])
print(d['ms_per_step'])
for k in d['roofline']['kernels']: print(k)
PY
