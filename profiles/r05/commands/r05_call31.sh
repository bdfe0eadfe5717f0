export TMPDIR=/tmp
for B in 32 64; do python tools/step_ab.py f16x3 $B -1 196 2>&1 | grep -v amdgpu.ids; done
for B in 32 64; do python tools/step_ab.py bf16 $B -1 196 2>&1 | grep -v amdgpu.ids; done
