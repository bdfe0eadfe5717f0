"""GPU: ms per DDPM step of the hipGraph loop over a grid of clip lengths and batch sizes, to look for cliffs (a kernel instantiation that
spills, a selection rule that falls off a tile shape):  shape_sweep.py [prec] [arch_mdm_l|arch_mdm]   -> us per clip and step, per shape"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oakink2-tamf_amd")]
sys.path.insert(0, os.path.join(ROOT, "tools")); import _ablib  # noqa: E702,F401
import torch
from oakink2_tamf_amd.hip_backend import TamfContext
from oracle import mdm_oracle as O

prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
aname = sys.argv[2] if len(sys.argv) > 2 else "arch_mdm_l"
oarch = {"arch_mdm_l": O.ARCH_MDM_L, "arch_mdm": O.ARCH_MDM}[aname]
arch = dict(latent_dim=oarch.latent_dim, ff_size=oarch.ff_size, num_layers=oarch.num_layers, num_heads=oarch.num_heads)
sd = O.det_state_dict(oarch, tag="sweep/w")
N = 50
tab = O.make_tables(N, "cosine")
ctx = TamfContext(arch, 1, 1, precision=prec)
ctx.load_state_dict(sd)
ctx.set_schedule(tab.posterior_mean_coef1, tab.posterior_mean_coef2, tab.posterior_log_variance_clipped)
Ts = [16, 32, 64, 96, 120, 123, 124, 139, 144, 160, 171, 176, 187, 196, 203, 204, 219, 224, 250]
print(f"{os.environ.get('TAMF_LIB_OVERRIDE', 'default').split('/')[-1]} {prec} {aname}: us per clip and DDPM step (ms per step)")
for B in (1, 8, 32, 64):
    row = []
    for T in Ts:
        ctx.resize(B, T) if hasattr(ctx, "resize") else None
        cond = O.det_cond(B, T, tag="x", arch=oarch)
        cd = {k: (v.cuda() if hasattr(v, "cuda") else v) for k, v in cond.items()}
        ctx.set_cond(cd["text_embedding"], cd["hand_side"], cd["shape"], cd["obj_embedding"], cd["obj_traj"])
        out = torch.empty(B, 99, 1, T, device="cuda")
        ctx.sample_loop(seed=1, out=out); torch.cuda.synchronize()
        t = time.perf_counter(); ctx.sample_loop(seed=2, out=out); ctx.sample_loop(seed=3, out=out); torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / (2 * N) * 1e3
        row.append(f"T{T}:{ms * 1e3 / B:6.1f}({ms:.2f})")
    print(f"B={B:3d} " + " ".join(row), flush=True)
