#!/bin/bash
# GPU: PMC counters of the attention kernel (one counter group per pass; no trace domains besides kernel-trace)
export TMPDIR=/tmp
p=${1:-bf16x3}
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" \
           "SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_INST_CYCLES_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAVES" \
           "GRBM_GUI_ACTIVE SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT"; do
  i=$((i+1))
  rm -rf gpurun_out/apmc
  timeout 150 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/apmc -o r -- python3 tools/attn_one.py $p 3 > /dev/null 2>&1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/apmc/**/*counter_collection.csv", recursive=True)
if not f: print("pass $i: no counters ($grp)"); raise SystemExit
acc = collections.OrderedDict()
for r in csv.DictReader(open(f[0])):
    if "attn_" not in r["Kernel_Name"]: continue
    acc.setdefault(r["Counter_Name"], {}).setdefault(r["Dispatch_Id"], 0.0)
    acc[r["Counter_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
print("pass $i", {k: list(v.values())[-1] for k, v in acc.items()})
PY
done
rm -rf gpurun_out/apmc
