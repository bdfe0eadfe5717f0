#!/bin/bash
# experiment: residual GEMMs of a few clips on 32 x 64 tiles (4 waves, six stages; selection bit 32) against 32 x 128 (8 waves, four stages)
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
{
python3 - <<'PY'
import sys, os
sys.path[:0] = [os.getcwd(), os.path.join(os.getcwd(), "oakink2-tamf_amd")]
import torch
from oakink2_tamf_amd.hip_backend import TamfContext, lib
from oracle import mdm_oracle as O
arch = dict(latent_dim=512, ff_size=2048, num_layers=8, num_heads=4)
sd = O.det_state_dict(O.ARCH_MDM_L, tag="x/w")
for prec in ("f16x3", "f32", "bf16"):
    for B, T in ((1, 160), (3, 196), (7, 123)):
        ctx = TamfContext(arch, B, T, precision=prec); ctx.load_state_dict(sd)
        c = O.det_cond(B, T, tag="c", arch=O.ARCH_MDM_L)
        cd = {k: (v.cuda() if hasattr(v, "cuda") else v) for k, v in c.items()}
        ctx.set_cond(cd["text_embedding"], cd["hand_side"], cd["shape"], cd["obj_embedding"], cd["obj_traj"])
        g = torch.Generator().manual_seed(5); x = torch.randn(B, 99, 1, T, generator=g).cuda(); t = torch.randint(0, 1000, (B,), generator=g).cuda()
        a = ctx.denoise(x, t).cpu()
        lib().tamf_set_gemm_tuning(0x20fffff); b = ctx.denoise(x, t).cpu(); lib().tamf_set_gemm_tuning(-1)
        print(prec, B, T, "same bits" if torch.equal(a, b) else "DIFFERENT", float((a - b).abs().max()))
        ctx.close()
PY
for p in f16x3 bf16 f32; do python tools/step_ab.py $p 1 0x20fffff,-1,0x20fffff,-1 160 2>&1 | grep -v amdgpu.ids; done
python tools/step_ab.py f16x3 4 0x20fffff,-1,0x20fffff,-1 160 2>&1 | grep -v amdgpu.ids
python tools/step_ab.py f16x3 8 0x20fffff,-1,0x20fffff,-1 160 2>&1 | grep -v amdgpu.ids
} > gpurun_out/r05/small_batch_32x64_c37.txt 2>&1
cut -c1-250 gpurun_out/r05/small_batch_32x64_c37.txt
