#!/bin/bash
# GPU: usage  pmc_generic.sh <kernel-substring> "<grp1>|<grp2>|..." -- python3 prog args...   (one counter group per pass)
export TMPDIR=/tmp
kname=$1; groups=$2; shift 3
IFS='|' read -ra G <<< "$groups"
i=0
for grp in "${G[@]}"; do
  i=$((i+1))
  rm -rf gpurun_out/gpmc
  timeout 200 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/gpmc -o r -- "$@" > gpurun_out/gpmc.log 2>&1
  python3 - "$kname" "$i" "$grp" <<'PY'
import csv, glob, collections, sys
kname, i, grp = sys.argv[1:4]
f = glob.glob("gpurun_out/gpmc/**/*counter_collection.csv", recursive=True)
if not f:
    print(f"pass {i}: no counters ({grp})"); print(open("gpurun_out/gpmc.log").read()[-400:]); raise SystemExit
acc = collections.OrderedDict()
for r in csv.DictReader(open(f[0])):
    if kname not in r["Kernel_Name"]: continue
    acc.setdefault(r["Counter_Name"], collections.OrderedDict()).setdefault(r["Dispatch_Id"], 0.0)
    acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
print(f"pass {i}", {k: list(v.values())[-1] for k, v in acc.items()})
PY
done
rm -rf gpurun_out/gpmc gpurun_out/gpmc.log
