#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
bash tools/energy_model.sh 5 > gpurun_out/r04/energy_run.log 2>&1
grep CASE gpurun_out/r04/energy_cases.txt | cut -c1-100
bash tools/pmc_step_totals.sh f16x3 > gpurun_out/r04/pmc_totals_f16x3.log 2>&1
bash tools/pmc_step_totals.sh bf16 > gpurun_out/r04/pmc_totals_bf16.log 2>&1
tail -40 gpurun_out/r04/pmc_totals_f16x3.log
