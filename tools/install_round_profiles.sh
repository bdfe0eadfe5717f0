#!/bin/bash
# Build container: move what tools/round_profiles_all.sh (+ the call script around it) left under gpurun_out/ into profiles/<round>/,
# replacing the files of the previous collection:   tools/install_round_profiles.sh <round> <old sha> <new sha> <call tag, e.g. c36>
set -e
R=$1; O=$2; S=$3; C=$4
P=profiles/$R; G=gpurun_out/prof
git rm -q --ignore-unmatch $P/bench_T160_*_$O.json $P/bench_b1_t160_$O.json $P/bench_config3_shard_$O.json $P/bench_config4_$O.json $P/bench_config5_shard_$O.json \
  $P/bench_default_$O.log $P/parity_report_$O.txt $P/pmc_attention_*_$O.txt $P/pmc_gemm_ffn1_f16x3_$O.txt $P/pmc_gemms_bf16_$O.txt \
  $P/rocprofv3_kernel_stats_*_$O.csv $P/stress_and_loop_errors_$O.txt $P/gpu_tests_full_$O.log
for d in f16x3 f32 bf16; do tail -n 1 $G/bench_T160_$d.log > $P/bench_T160_${d}_$S.json; done
tail -n 1 $G/bench_b1_t160.json > $P/bench_b1_t160_$S.json
tail -n 1 $G/bench_config3_shard.log > $P/bench_config3_shard_$S.json
tail -n 1 $G/bench_config5_shard.log > $P/bench_config5_shard_$S.json
tail -n 1 $G/bench_config4.json > $P/bench_config4_$S.json
grep -v amdgpu.ids $G/bench_default.log > $P/bench_default_$S.log
grep -v amdgpu.ids $G/parity_report.txt > $P/parity_report_$S.txt
for d in f16x3 bf16; do grep -v amdgpu.ids $G/pmc/pmc_attention_$d.txt > $P/pmc_attention_${d}_$S.txt; done
grep -v amdgpu.ids $G/pmc/pmc_gemm_ffn1_f16x3.txt > $P/pmc_gemm_ffn1_f16x3_$S.txt
grep -v amdgpu.ids $G/pmc/pmc_gemms_bf16.txt > $P/pmc_gemms_bf16_$S.txt
for d in f16x3 f32 bf16 bf16x3; do
  cp $G/kernel_stats_$d.csv $P/rocprofv3_kernel_stats_${d}_B64_T196_100steps_$S.csv
  cp $G/hbm_traffic_$d.json $G/pmc_step_totals_$d.json $P/
  cp $G/step_profile_$d.json $P/step_profile_hipevents_$d.json
done
cp gpurun_out/$R/stress_and_loop_errors_$C.txt $P/stress_and_loop_errors_$S.txt
cp gpurun_out/$R/gpu_tests_full_$C.log $P/gpu_tests_full_$S.log
[ -f gpurun_out/$R/shape_sweep_f32_$C.txt ] && { git rm -q --ignore-unmatch $P/shape_sweep_f32_c*.txt; cp gpurun_out/$R/shape_sweep_f32_$C.txt $P/shape_sweep_f32_$C.txt; }
python tools/design_table.py $P > $P/kernel_table.md
python3 - $P $S <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
P, S = sys.argv[1] + "/", sys.argv[2]
last = lambda f: json.loads([l for l in open(P + f) if l.startswith("{")][-1])
for m in ("f16x3", "f32", "bf16", "bf16x3"):
    d = json.load(open(P + f"hbm_traffic_{m}.json"))
    print(m, d["commit"], "digest ok" if d["csrc_sha16"] == bench.csrc_digest() else "DIGEST MISMATCH")
d = last(f"bench_default_{S}.log")
r = d["roofline"]
print("default", round(d["value"], 1), "frames/s", round(d["ms_per_step"] / 1e3, 4), "ms/step;", "f32", round(r["f32_value"], 1), round(r["f32_ms_per_ddpm_step"], 4),
      round(r["f32_whole_path_frac"], 4), "| frac", round(r["frac"], 4), "f32_frac", round(r["f32_frac"], 4), "attn", round(r["attention_frac"], 4), round(r["f32_attention_frac"], 4),
      "| sustained", round(r["mfma_sustained_tflops"]), "stale", r["traffic_stale"])
print("check", d["check"]["max_abs_err_vs_oracle"])
for k, v in d["other_dtypes"].items():
    print(" ", k, round(v["value"], 1), round(v["ms_per_ddpm_step"], 4), round(v["whole_path_frac_of_peak"], 4), round((v.get("roofline") or {}).get("frac", 0), 4), round(v.get("whole_path_tflops", 0), 1))
print("power", round(d["power"]["watts"]), round(d["power"]["sclk_mhz"]), "cpu", round(d["cpu_baseline"]["value"], 2))
for f in ("bench_config4", "bench_b1_t160", "bench_config3_shard", "bench_config5_shard", "bench_T160_f16x3", "bench_T160_f32", "bench_T160_bf16"):
    x = last(f"{f}_{S}.json")
    print(f, round(x["value"], 1), round(x["ms_per_step"], 3), x["dtype"], x["roofline"].get("f32_value"))
c4 = last(f"bench_config4_{S}.json")
print({k: round(v["ms_per_batch"], 3) for k, v in c4["other_dtypes"].items()}, c4["cpu_baseline"]["sample"][-30:])
PY
