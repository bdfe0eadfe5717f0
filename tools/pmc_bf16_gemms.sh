G="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY|SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS|SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE|GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM|SQ_LDS_ADDR_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_UNALIGNED_STALL"
echo "== bf16 FFN1 clip"; bash tools/pmc_generic.sh clip_gemm_kernel "$G" -- python3 tools/kbench_one.py bf16 0 -1 13312 2048 512 5
echo "== bf16 FFN2 LN tile"; bash tools/pmc_generic.sh gemm_kernel "$G" -- python3 tools/kbench_one.py bf16 2 -1 13312 512 2048 5
echo "== bf16 QKV"; bash tools/pmc_generic.sh gemm_kernel "$G" -- python3 tools/kbench_one.py bf16 1 -1 13312 1536 512 5
