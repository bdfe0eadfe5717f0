set -x
bash tools/collect_round_profiles.sh "f16x3 f32 bf16 bf16x3" > gpurun_out/prof_collect.log 2>&1
mkdir -p gpurun_out/prof/pmc
for p in f16x3 bf16 f32 bf16x3; do bash tools/pmc_step_totals.sh $p gpurun_out/prof/pmc_step_totals_$p.json > gpurun_out/prof/pmc/pmc_step_totals_$p.log 2>&1; done
for p in f16x3 bf16; do bash tools/attn_pmc.sh $p > gpurun_out/prof/pmc/pmc_attention_$p.txt 2>&1; done
bash tools/pmc_bf16_gemms.sh > gpurun_out/prof/pmc/pmc_gemms_bf16.txt 2>&1
bash tools/pmc_generic.sh clip_gemm_kernel "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY|SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS|SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE|GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM|TCC_HIT_sum TCC_MISS_sum|FETCH_SIZE|WRITE_SIZE" -- python3 tools/kbench_one.py f16x3 10 -1 13312 2048 512 5 > gpurun_out/prof/pmc/pmc_gemm_ffn1_f16x3.txt 2>&1
python tests/scripts/parity_report.py > gpurun_out/prof/parity_report.txt 2>&1
ls -la gpurun_out/prof gpurun_out/prof/pmc
tail -8 gpurun_out/prof_collect.log
