#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
V="-1,0x8fffff,0x20fffff,0x40fffff,0x80fffff,0x480fffff,0x100fffff,0x10fffff,0x400fffff,0x200fffff,0x4fffff,-1"
{ for p in f16x3 bf16 f32; do python tools/step_ab.py $p 64 $V 160; python tools/step_ab.py $p 32 $V 196; done; } 2>&1 | grep -E "variant" > gpurun_out/r04/sel_sweep_t160_b32_c19.txt
python - <<'PY'
import re
for l in open('gpurun_out/r04/sel_sweep_t160_b32_c19.txt'):
    m=re.match(r'(\S+) B=(\d+) T=(\d+) variant (\S+): step (\d+) us \| (.*)', l)
    if m: print(m.group(1), m.group(2), m.group(3), m.group(4), m.group(5), ' '.join(x for x in m.group(6).split() if not x.startswith(('gemm_input','gemm_head'))))
PY
