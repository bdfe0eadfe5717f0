#!/bin/bash
# Re-fetch, built: FFN1's two rounds split by columns (selection bit 32) against the default clip-split rounds - bits, loop time, FETCH / WRITE of FFN1
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 600 python -m pytest tests/test_hip_fullsize.py -q -m gpu -k "column_split or kernel_selections_give_the_same_bits" 2>&1 | tail -4
{
for rep in 1 2 3; do
for p in f16x3 bf16 f32; do
  timeout 300 python tools/loop_time.py $p 64 200 3 -1 2>&1 | grep ms/step
  timeout 300 python tools/loop_time.py $p 64 200 3 0x20fffff 2>&1 | grep ms/step
done
done
for p in f16x3 bf16 f32; do
  echo "== $p FFN1 default order"
  bash tools/pmc_generic.sh EpiBiasAct "FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum TCC_MISS_sum" -- python3 tools/loop_time.py $p 64 10 1 -1
  echo "== $p FFN1 column-split rounds"
  bash tools/pmc_generic.sh EpiBiasAct "FETCH_SIZE|WRITE_SIZE|TCC_HIT_sum TCC_MISS_sum" -- python3 tools/loop_time.py $p 64 10 1 0x20fffff
done
} > gpurun_out/r06/ab_colsplit_c07.txt 2>&1
cat gpurun_out/r06/ab_colsplit_c07.txt
