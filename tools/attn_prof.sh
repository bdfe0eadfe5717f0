#!/bin/bash
# GPU: per-launch duration of the attention kernel for several waves-per-workgroup settings (rocprofv3 kernel trace)
export TMPDIR=/tmp
for nw in ${1:-8 4}; do
  export TAMF_ATTN_NW=$nw
  for p in ${2:-bf16x3 bf16}; do
    rm -rf gpurun_out/ap
    timeout 150 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/ap -o r -- python3 tools/attn_one.py $p 5 > /dev/null 2>&1
    python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/ap/**/*kernel_trace.csv", recursive=True)
if not f: print("NW=$nw $p: no trace"); raise SystemExit
d = [ (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in csv.DictReader(open(f[0])) if "attn_kernel" in r["Kernel_Name"]]
print("NW=$nw $p attn_kernel us:", [round(x, 1) for x in d])
PY
  done
done
rm -rf gpurun_out/ap
