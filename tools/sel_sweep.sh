# GPU: whole-loop ms per DDPM step under each single kernel-selection bit of tamf_set_gemm_tuning (defaults first and last: drift of the box)
for p in f16x3 bf16; do
  for sel in 0 1 2 4 8 64 512 1024 0; do
    t=$(printf "0x%xfffff" $sel); [ $sel = 0 ] && t=-1
    python tools/loop_time.py $p 64 100 2 $t 2>&1 | grep ms/step
  done
done
