#!/bin/bash
# launcher smoke with respacing: synthetic clips, 1000 steps against 100 of 1000 (own flag --timestep_respacing), wall time of each
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
export PYTHONPATH=$PWD/oakink2-tamf_amd
{
cd gpurun_out/r06 && mkdir -p cli && cd cli
( time python -m oakink2_tamf_amd.launch.sample --cfg ../../../config/arch_mdm_l.yml --synthetic 64,196 --debug.sample_save_offset t/full --commit ) 2>&1 | grep -v amdgpu.ids | tail -6
( time python -m oakink2_tamf_amd.launch.sample --cfg ../../../config/arch_mdm_l.yml --synthetic 64,196 --timestep_respacing 100 --debug.sample_save_offset t/r100 --commit ) 2>&1 | grep -v amdgpu.ids | tail -6
python - <<'PY'
import numpy as np, glob
for d in ("full", "r100"):
    fs = sorted(glob.glob(f"common/sample/main/sample/t/{d}/*.npy"))
    a = np.stack([np.load(f) for f in fs])
    print(d, len(fs), a.shape, a.dtype, "finite", bool(np.isfinite(a).all()), "mean |x|", float(np.abs(a).mean()))
PY
} > gpurun_out/r06/cli_respacing_c16.txt 2>&1
cd ../../.. && rm -rf gpurun_out/r06/cli
cat gpurun_out/r06/cli_respacing_c16.txt
