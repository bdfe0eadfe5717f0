#!/bin/bash
# (1) can dependent launches of one stream overlap on gfx950 (any-order launch + device-side ready flags)?  placement of workgroups on XCDs
# (2) same-box library yardstick: torch F.linear / SDPA at the step's shapes against the hand-written launches
# (3) bench.py defaults with the torch_rocm_baseline leg
export TMPDIR=/tmp
mkdir -p gpurun_out/r06
hipcc -O3 --offload-arch=gfx950 -o gpurun_out/anyorder tools/micro/anyorder.hip 2> gpurun_out/r06/anyorder_build.log
timeout 120 ./gpurun_out/anyorder > gpurun_out/r06/anyorder_c01.txt 2>&1
cat gpurun_out/r06/anyorder_c01.txt
timeout 600 python tools/lib_yardstick.py gpurun_out/r06/lib_yardstick_c01.json > gpurun_out/r06/lib_yardstick_c01.txt 2>&1
tail -40 gpurun_out/r06/lib_yardstick_c01.txt
timeout 900 python bench.py > gpurun_out/r06/bench_default_c01.log 2> gpurun_out/r06/bench_default_c01.err
python - <<'PY'
import json
l = json.loads(open("gpurun_out/r06/bench_default_c01.log").read().strip().splitlines()[-1])
print({k: l[k] for k in ("value", "ms_per_ddpm_step", "check_ok")}, l["check"])
print(json.dumps(l.get("torch_rocm_baseline"), indent=1)[:3000])
PY
tail -5 gpurun_out/r06/bench_default_c01.err
