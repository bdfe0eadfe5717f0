#!/bin/bash
# GPU: rocprofv3 kernel-trace summary of the hipGraph loop for the A build (lib/libtamf_hip_A.so, tools/ab_build.sh) and the
# working-tree build on the same box:  tools/ab_kstats.sh [prec] [B]
export TMPDIR=/tmp
prec=${1:-f16x3}; B=${2:-64}
A=$PWD/oakink2-tamf_amd/oakink2_tamf_amd/lib/libtamf_hip_A.so
out=gpurun_out/abk; rm -rf $out; mkdir -p $out
for which in A B; do
  if [ $which = A ]; then export TAMF_LIB_OVERRIDE=$A; else unset TAMF_LIB_OVERRIDE; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$which -o r -- python3 tools/loop_time.py $prec $B 100 2 > $out/$which.log 2>&1
  python3 - $out $which <<'PY'
import csv, glob, sys, re
out, which = sys.argv[1:3]
f = glob.glob(f"{out}/{which}/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
print(f"--- build {which}")
for r in rows[:14]:
    name = re.sub(r"\(.*", "", r["Name"])[:110]
    print("%-110s calls %6s avg %8.2f us  total %9.3f ms" % (name, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
