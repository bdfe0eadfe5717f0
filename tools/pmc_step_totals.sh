#!/bin/bash
# GPU: hardware-counter totals of ONE DDPM step (all kernels of the loop), for the energy model:  pmc_step_totals.sh <dtype> [out.json]
# One rocprofv3 --pmc pass per counter group over `bench.py --ddpm-steps 10`; every counter is summed over all dispatches of the step's
# kernels and divided by the number of steps that ran (= attention dispatches / 8 layers).
export TMPDIR=/tmp
dt=${1:-f16x3}; out=${2:-gpurun_out/prof/pmc_step_totals_$dt.json}
tmpd=$(mktemp -d /tmp/pmc_tot_XXXXXX)  # (fixed /tmp/pmc_tot_*.json names let a pass of another mode leak into the merge: ADVICE r4)
groups="SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU|SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES|SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS|TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum|TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum|GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT"
IFS='|' read -ra G <<< "$groups"
i=0
for grp in "${G[@]}"; do
  i=$((i+1))
  rm -rf gpurun_out/gpmc
  timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d gpurun_out/gpmc -o r -- python3 bench.py --no-power --steps 1 --warmup 0 --ddpm-steps 10 --no-cpu-baseline --also "" --fp32-loops 0 --check-clips 0 --dtype $dt > gpurun_out/gpmc.log 2>&1
  python3 - $i $tmpd <<'PY' || { echo "pmc_step_totals.sh: pass $i ($grp) produced no counters - not merging a partial total" >&2; rm -rf $tmpd; exit 1; }
import csv, glob, collections, json, sys
f = glob.glob("gpurun_out/gpmc/**/*counter_collection.csv", recursive=True)
if not f:
    print("no counters in pass", sys.argv[1]); print(open("gpurun_out/gpmc.log").read()[-300:]); raise SystemExit(1)
tot = collections.defaultdict(float); attn = set()
for r in csv.DictReader(open(f[0])):
    k = r["Kernel_Name"]
    if not any(s in k for s in ("gemm_kernel", "clip_gemm_kernel", "attn_")): continue
    if "OpF32" in k and "gemm_kernel<OpF32" in k and sys.argv[1] and False: continue
    tot[r["Counter_Name"]] += float(r["Counter_Value"])
    if "attn_" in k: attn.add(r["Dispatch_Id"])
steps = len(attn) / 8.0
json.dump({"steps": steps, "per_step": {k: v / steps for k, v in tot.items()}}, open(f"{sys.argv[2]}/pass_{int(sys.argv[1]):02d}.json", "w"))
print("pass", sys.argv[1], "steps", steps, {k: round(v / steps) for k, v in tot.items()})
PY
done
python3 - $dt $out $tmpd <<'PY'
import glob, json, sys
m = {}
steps = None
passes = sorted(glob.glob(sys.argv[3] + "/pass_*.json"))
for f in passes:
    d = json.load(open(f)); m.update(d["per_step"]); steps = d["steps"]
json.dump({"dtype": sys.argv[1], "B": 64, "T": 196, "passes_merged": len(passes), "what": "rocprofv3 --pmc totals over all GEMM / attention / LayerNorm dispatches of bench.py --ddpm-steps 10, per DDPM step", "steps_counted": steps, "per_step": m}, open(sys.argv[2], "w"), indent=1)
print(open(sys.argv[2]).read())
PY
rm -rf gpurun_out/gpmc gpurun_out/gpmc.log $tmpd
