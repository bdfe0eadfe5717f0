"""GPU: the 1000-step loop of every arithmetic mode for about `secs` seconds each, with wall-clock stamps for tools/power_sampler.py:
   python tools/energy_loops.py [secs] [B] [T]   ->  lines "CASE loop_<dtype> t0 t1 <DDPM steps per second> step" """
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oakink2-tamf_amd")]
import torch
from oakink2_tamf_amd.hip_backend import TamfContext
from oracle import mdm_oracle as O
secs = float(sys.argv[1]) if len(sys.argv) > 1 else 8.0
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
T = int(sys.argv[3]) if len(sys.argv) > 3 else 196
N = 200
arch = dict(latent_dim=512, ff_size=2048, num_layers=8, num_heads=4)
sd = O.det_state_dict(O.ARCH_MDM_L, tag="bench/w")
tab = O.make_tables(N, "cosine")
cond = O.det_cond(B, T, tag="x", arch=O.ARCH_MDM_L)
cd = {k: (v.cuda() if hasattr(v, "cuda") else v) for k, v in cond.items()}
out = torch.empty(B, 99, 1, T, device="cuda")
for prec in ("f16x3", "bf16x3", "bf16", "f32"):
    ctx = TamfContext(arch, B, T, precision=prec)
    ctx.load_state_dict(sd)
    ctx.set_schedule(tab.posterior_mean_coef1, tab.posterior_mean_coef2, tab.posterior_log_variance_clipped)
    ctx.set_cond(cd["text_embedding"], cd["hand_side"], cd["shape"], cd["obj_embedding"], cd["obj_traj"])
    ctx.sample_loop(seed=1, out=out); torch.cuda.synchronize()
    t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        ctx.sample_loop(seed=2 + n, out=out); torch.cuda.synchronize(); n += 1
    t1 = time.time()
    print(f"CASE loop_{prec} {t0:.3f} {t1:.3f} {n * N / (t1 - t0):.6e} step", flush=True)
    ctx.close()
