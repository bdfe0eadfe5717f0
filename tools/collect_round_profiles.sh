#!/bin/bash
# GPU: regenerate the per-round profile artefacts under gpurun_out/prof (copy what should be judged into profiles/rNN):
#   kernel_stats_<dtype>.csv       rocprofv3 --kernel-trace --stats summary of a 100-step bench run
#   hbm_traffic_<dtype>.json       FETCH_SIZE / WRITE_SIZE (separate --pmc passes), per kernel class and launch
#   step_profile_<dtype>.json      bench.py's in-run HIP-event profile of one step
#   bench_default.log              the default bench line
# Pass the commit the tree was built from as TAMF_COMMIT=<sha> (the GPU box has no .git): it is stamped into hbm_traffic_*.json,
# which bench.py reports as roofline.traffic_source.
export TMPDIR=/tmp
# (an unstamped or default-round run used to overwrite the previous round's evidence)
if [ -z "$TAMF_ROUND" ] || [ -z "$TAMF_COMMIT" ]; then echo "collect_round_profiles.sh: set TAMF_ROUND (e.g. r05) and TAMF_COMMIT (the sha the tree was built from)" >&2; exit 2; fi
out=gpurun_out/prof; rm -rf $out; mkdir -p $out
for dt in ${1:-f16x3 f32 bf16 bf16x3}; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ks_$dt -o r -- python3 bench.py --no-power --steps 1 --warmup 0 --ddpm-steps 100 --no-cpu-baseline --also "" --fp32-loops 0 --dtype $dt > $out/ks_$dt.log 2>&1
  python3 - $out $dt <<'PY'
import csv, glob, sys
out, dt = sys.argv[1:3]
f = glob.glob(f"{out}/ks_{dt}/**/*kernel_stats.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
with open(f"{out}/kernel_stats_{dt}.csv", "w") as o:
    o.write(f"# rocprofv3 --kernel-trace --stats -- python3 bench.py --no-power --steps 1 --warmup 0 --ddpm-steps 100 --no-cpu-baseline --also \"\" --fp32-loops 0 --dtype {dt}\n")
    o.write("kernel,calls,total_ms,avg_us,percent\n")
    for r in rows:
        o.write('"%s",%s,%.3f,%.2f,%s\n' % (r["Name"], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, r["Percentage"]))
PY
  rm -rf $out/ks_$dt
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 600 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $out/pmc_${dt}_$ctr -o r -- python3 bench.py --no-power --steps 1 --warmup 0 --ddpm-steps 10 --no-cpu-baseline --also "" --fp32-loops 0 --dtype $dt > $out/pmc_$dt.log 2>&1
  done
  python3 - $out $dt <<'PY'
import csv, glob, json, os, sys, collections
sys.path.insert(0, os.getcwd())
out, dt = sys.argv[1:3]
CLASSES = [("EpiQKV", "gemm_qkv"), ("EpiQK<", "gemm_qk"), ("EpiVt", "gemm_v"), ("attn_", "attention"), ("EpiBiasAct", "gemm_ffn1_gelu"),
           ("EpiResid", "pair:gemm_outproj:gemm_ffn2"),     # the residual GEMMs (deferred LayerNorm): out-proj and FFN2 alternate, layer by layer
           ("EpiSeqRows", "gemm_input_merge2"), ("EpiHead", "gemm_head_ddpm")]
def per_launch(ctr):
    f = glob.glob(f"{out}/pmc_{dt}_{ctr}/**/*counter_collection.csv", recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    rows = sorted(csv.DictReader(open(f[0])), key=lambda r: int(r["Dispatch_Id"]))
    pair_seen = collections.defaultdict(dict)
    for r in rows:
        for pat, name in CLASSES:
            if pat in r["Kernel_Name"]:
                if name.startswith("pair:"):
                    seen = pair_seen[pat]
                    if r["Dispatch_Id"] not in seen:
                        seen[r["Dispatch_Id"]] = len(seen)
                    name = name.split(":")[1 + seen[r["Dispatch_Id"]] % 2]
                acc[name][r["Dispatch_Id"]] += float(r["Counter_Value"])
                break
    return {k: sorted(v.values())[len(v) // 2] for k, v in acc.items()}  # median over launches (merge0's BiasAct launch is the minority)
fe, wr = per_launch("FETCH_SIZE"), per_launch("WRITE_SIZE")
kern = {}
for k in sorted(set(fe) | set(wr)):
    name = k
    f_b, w_b = fe.get(k, 0.0) * 1024.0 * 2.0, wr.get(k, 0.0) * 1024.0
    kern[name] = {"fetch_bytes_corrected": f_b, "write_bytes": w_b, "traffic_bytes_per_launch": f_b + w_b}
json.dump({"command": "rocprofv3 --pmc FETCH_SIZE (and, separately, --pmc WRITE_SIZE) --kernel-trace -- python3 bench.py --no-power --steps 1 --warmup 0 --ddpm-steps 10 --no-cpu-baseline --also '' --fp32-loops 0 --dtype " + dt,
           "correction": "counters are reported in KiB; gfx950: FETCH_SIZE reports 1/2 of the bytes of wide (16 B/lane) coalesced reads -> doubled (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact; counters are fabric-side (Infinity Cache hits included); median over the launches of a kernel class",
           "dtype": dt, "B": 64, "T": 196, "commit": os.environ.get("TAMF_COMMIT", "unstamped"), "csrc_sha16": __import__("bench").csrc_digest(), "kernels": kern}, open(f"{out}/hbm_traffic_{dt}.json", "w"), indent=1)
PY
  rm -rf $out/pmc_${dt}_FETCH_SIZE $out/pmc_${dt}_WRITE_SIZE
  python3 bench.py --no-power --steps 1 --warmup 1 --no-cpu-baseline --also "" --fp32-loops 0 --dtype $dt --profile-out $out/step_profile_$dt.json > $out/step_$dt.log 2>&1
done
# the traffic files of THIS tree go where bench.py looks for them (roofline.traffic / traffic_stale) before the bench lines are taken
mkdir -p profiles/$TAMF_ROUND && cp $out/hbm_traffic_*.json profiles/$TAMF_ROUND/
timeout 900 python3 bench.py > $out/bench_default.log 2>&1
tail -n 1 $out/bench_default.log | cut -c1-600
# the clip length the reference's dataset emits (T = 160) and the two 8-GPU presets' per-GPU shards, one line each
for dt in f16x3 f32 bf16; do
  timeout 600 python3 bench.py --frames 160 --dtype $dt --also "" --fp32-loops 0 --no-cpu-baseline --steps 2 --warmup 1 > $out/bench_T160_$dt.log 2>&1
  tail -n 1 $out/bench_T160_$dt.log | cut -c1-200
done
timeout 600 python3 bench.py --config 3 --also "" --fp32-loops 0 --no-cpu-baseline --steps 2 --warmup 1 > $out/bench_config3_shard.log 2>&1
timeout 600 python3 bench.py --config 5 --also "" --fp32-loops 0 --no-cpu-baseline --steps 2 --warmup 1 > $out/bench_config5_shard.log 2>&1
tail -n 1 $out/bench_config3_shard.log | cut -c1-200; tail -n 1 $out/bench_config5_shard.log | cut -c1-200
ls $out
