"""GPU: per-kernel HIP-event profile of one denoiser step under different GEMM-selection bits (tamf_set_gemm_tuning):
   python tools/step_ab.py [prec] [B] [variants, comma list of ints; -1 = default] [T]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oakink2-tamf_amd")]
sys.path.insert(0, os.path.join(ROOT, "tools")); import _ablib  # noqa: E702,F401  (TAMF_LIB_OVERRIDE: A/B builds)
import torch
from oakink2_tamf_amd.hip_backend import TamfContext, lib
from oracle import mdm_oracle as O

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
variants = [int(v, 0) for v in sys.argv[3].split(",")] if len(sys.argv) > 3 else [-1]
T = int(sys.argv[4]) if len(sys.argv) > 4 else 196
arch = dict(latent_dim=512, ff_size=2048, num_layers=8, num_heads=4)
sd = O.det_state_dict(O.ARCH_MDM_L, tag="bench/w")
tab = O.make_tables(1000, "cosine")
ctx = TamfContext(arch, B, T, precision=prec)
ctx.load_state_dict(sd)
ctx.set_schedule(tab.posterior_mean_coef1, tab.posterior_mean_coef2, tab.posterior_log_variance_clipped)
cond = O.det_cond(B, T, tag="x", arch=O.ARCH_MDM_L)
cd = {k: (v.cuda() if hasattr(v, "cuda") else v) for k, v in cond.items()}
ctx.set_cond(cd["text_embedding"], cd["hand_side"], cd["shape"], cd["obj_embedding"], cd["obj_traj"])
for v in variants:
    lib().tamf_set_gemm_tuning(v)
    agg = {}
    for r in range(6):
        rows = ctx.step_profile()
        if r == 0:
            continue
        for n, ms, fl in rows:
            a = agg.setdefault(n, [0.0, 0])
            a[0] += ms; a[1] += 1
    tot = sum(a[0] for a in agg.values()) / 5
    print(f"{prec} B={B} T={T} variant {v:#x}: step {tot*1e3:.0f} us | " + " ".join(f"{n}={a[0]/a[1]*1e3:.1f}" for n, a in agg.items()), flush=True)
lib().tamf_set_gemm_tuning(-1)
