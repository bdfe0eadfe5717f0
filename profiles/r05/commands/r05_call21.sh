#!/bin/bash
# config 4: kernel timeline of 30 forwards (rocprofv3 --kernel-trace): busy time, gaps, the conditioning kernels
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d /tmp/c4trace -o r -- python3 $GRAFT_REPO_ROOT/bench.py --config 4 --no-cpu-baseline --no-power --also "" --steps 30 --warmup 5 > /tmp/c4.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find /tmp/c4trace -name "*kernel_trace.csv" | head -1)
python3 - $f > gpurun_out/r05/config4_timeline_c21.txt <<'PY'
import csv, sys, collections
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# one forward = from a refine_in_kernel back to the kernel before the next hand_side_kernel; take the last 20 forwards
idx = [i for i, r in enumerate(rows) if "hand_side_kernel" in r[2]]
print("forwards seen", len(idx))
per = []
for a, b in zip(idx[-21:-1], idx[-20:]):
    seg = rows[a:b]
    busy = sum(e - s for s, e, _ in seg)
    span = rows[b][0] - seg[0][0]
    per.append((span, busy, len(seg), seg))
span = sum(p[0] for p in per) / len(per); busy = sum(p[1] for p in per) / len(per)
print(f"per forward: span {span/1e3:.1f} us (start of hand_side_kernel to the next one), kernels busy {busy/1e3:.1f} us, {per[0][2]} kernels")
seg = per[-1][3]; nxt = rows[idx[-1]][0]
for i, (s, e, n) in enumerate(seg):
    gap = (seg[i + 1][0] if i + 1 < len(seg) else nxt) - e
    print(f"{(s - seg[0][0])/1e3:9.1f} us  dur {(e - s)/1e3:7.1f}  gap after {gap/1e3:7.1f}  {n[:100]}")
PY
tail -3 /tmp/c4.log | cut -c1-300
head -70 gpurun_out/r05/config4_timeline_c21.txt | cut -c1-170
