#!/bin/bash
# conditioning precompute in two launches (prefix_rows_kernel, cobj_kernel): parity subsets, config 4 again, its timeline
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
{
python __graft_entry__.py smoke 2>&1 | tail -4
timeout 1500 python -m pytest tests/test_hip_forward.py tests/test_hip_robustness.py tests/test_hip_module.py tests/test_hip_pipeline.py -x -q -m gpu 2>&1 | tail -4
timeout 600 python bench.py --config 4 --no-cpu-baseline --also f32,bf16 2>/dev/null | tail -n 1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('config 4 ms/batch', d['ms_per_step'], 'check', d['check'], {k:v['ms_per_batch'] for k,v in d['other_dtypes'].items()})"
} > gpurun_out/r05/cond_two_launches_c22.txt 2>&1
cat gpurun_out/r05/cond_two_launches_c22.txt | grep -v amdgpu.ids
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/c4trace -o r -- python3 $GRAFT_REPO_ROOT/bench.py --config 4 --no-cpu-baseline --no-power --also "" --steps 30 --warmup 5 > /tmp/c4.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/c4trace -name "*kernel_trace.csv" | head -1)
python3 - $f > gpurun_out/r05/config4_timeline_c22.txt <<'PY'
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
idx = [i for i, r in enumerate(rows) if "prefix_rows_kernel" in r[2]]
print("forwards seen", len(idx))
per = []
for a, b in zip(idx[-21:-1], idx[-20:]):
    seg = rows[a:b]
    per.append((rows[b][0] - seg[0][0], sum(e - s for s, e, _ in seg), len(seg), seg))
print(f"per forward: span {sum(p[0] for p in per)/len(per)/1e3:.1f} us, kernels busy {sum(p[1] for p in per)/len(per)/1e3:.1f} us, {per[0][2]} kernels")
seg = per[-1][3]; nxt = rows[idx[-1]][0]
for i, (s, e, n) in enumerate(seg[:8]):
    print(f"{(s - seg[0][0])/1e3:9.1f} us  dur {(e - s)/1e3:7.1f}  {n[:100]}")
PY
head -12 gpurun_out/r05/config4_timeline_c22.txt | cut -c1-160
