#!/bin/bash
# one clip per call: the 1000-step loop as hipGraphs of 10 steps against plain launches (a 430-node graph launch may cost the host more than its 4 ms of GPU work)
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
{
for p in f16x3 bf16 f32; do
  for rep in 1 2; do
    python bench.py --batch 1 --frames 160 --dtype $p --also "" --fp32-loops 0 --no-cpu-baseline --no-power --steps 3 --warmup 1 2>/dev/null | tail -n 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$p graph   ', d['ms_per_step']/1000, 'ms/step')"
    python bench.py --batch 1 --frames 160 --dtype $p --also "" --fp32-loops 0 --no-cpu-baseline --no-power --steps 3 --warmup 1 --no-graph 2>/dev/null | tail -n 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$p no-graph', d['ms_per_step']/1000, 'ms/step')"
  done
done
for B in 4 8 16; do
  python bench.py --batch $B --frames 160 --dtype f16x3 --also "" --fp32-loops 0 --no-cpu-baseline --no-power --steps 3 --warmup 1 2>/dev/null | tail -n 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('f16x3 B=$B graph   ', d['ms_per_step']/1000, 'ms/step')"
  python bench.py --batch $B --frames 160 --dtype f16x3 --also "" --fp32-loops 0 --no-cpu-baseline --no-power --steps 3 --warmup 1 --no-graph 2>/dev/null | tail -n 1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('f16x3 B=$B no-graph', d['ms_per_step']/1000, 'ms/step')"
done
} > gpurun_out/r05/graph_vs_plain_small_c35.txt 2>&1
cat gpurun_out/r05/graph_vs_plain_small_c35.txt
