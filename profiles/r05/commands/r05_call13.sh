#!/bin/bash
# what does the hipGraph buy per evaluation?  the 1000-step loop with and without it (same process order, two rounds)
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
{
for rep in 1 2; do for dt in f16x3 bf16; do
for g in "" "--no-graph"; do
python bench.py --no-power --steps 1 --warmup 1 --ddpm-steps 200 --no-cpu-baseline --also "" --fp32-loops 0 --check-clips 0 --dtype $dt $g 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(j['dtype'], 'hipgraph', j['config']['hipgraph'], 'ms/step', round(j['ms_per_ddpm_step'],4))"
done; done; done
} > gpurun_out/r05/graph_vs_plain_c13.txt 2>&1
cat gpurun_out/r05/graph_vs_plain_c13.txt
