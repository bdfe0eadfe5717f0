#!/bin/bash
# VERDICT r5 #1b: SQ_VALU_MFMA_BUSY_CYCLES of the LIBRARY's kernels at the step's shapes, beside ours (pmc_gemms_bf16_*.txt): one counter group per pass
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
G="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS|GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY"
{
echo "(kernel substring '' = every kernel of the process; pmc_generic.sh prints the LAST dispatch = the timed call's kernel)"
for shape in "13312 2048 512" "13312 512 2048" "13312 1536 512" "13312 512 512"; do
  for dt in bf16 f32; do
    echo "== library F.linear $dt $shape"
    bash tools/pmc_generic.sh "" "$G" -- python3 tools/lib_one.py $dt $shape 3
  done
done
for dt in bf16 f16; do
  echo "== library SDPA $dt 64 4 201 128"
  bash tools/pmc_generic.sh "" "$G" -- python3 tools/lib_one.py $dt sdpa 64 4 201 128 3
done
echo "== kernel names the library dispatched (kernel trace of one bf16 FFN2 call and one bf16 SDPA call)"
rm -rf gpurun_out/kt; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -o r -- python3 tools/lib_one.py bf16 13312 512 2048 3 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/kt/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:4]: print(r["Name"][:150], r["Calls"], "%.1f us" % (float(r["AverageNs"]) / 1e3))
PY
rm -rf gpurun_out/kt; timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kt -o r -- python3 tools/lib_one.py bf16 sdpa 64 4 201 128 3 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("gpurun_out/kt/**/*kernel_stats.csv", recursive=True):
    for r in list(csv.DictReader(open(f)))[:4]: print(r["Name"][:150], r["Calls"], "%.1f us" % (float(r["AverageNs"]) / 1e3))
PY
rm -rf gpurun_out/kt
} > gpurun_out/r06/pmc_library_c15.txt 2>&1
cat gpurun_out/r06/pmc_library_c15.txt
