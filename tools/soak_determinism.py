"""GPU: soak test of the hipGraph loop - the same seed must give the same bits, loop after loop (a race in a counted wait or an LDS
stage reuse shows up as a rare difference):  python tools/soak_determinism.py [prec] [B] [T] [loops] [ddpm_steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oakink2-tamf_amd")]
import torch
from oakink2_tamf_amd.hip_backend import TamfContext
from oracle import mdm_oracle as O
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
T = int(sys.argv[3]) if len(sys.argv) > 3 else 196
loops = int(sys.argv[4]) if len(sys.argv) > 4 else 20
N = int(sys.argv[5]) if len(sys.argv) > 5 else 1000
arch = dict(latent_dim=512, ff_size=2048, num_layers=8, num_heads=4)
sd = O.det_state_dict(O.ARCH_MDM_L, tag="bench/w")
tab = O.make_tables(N, "cosine")
ctx = TamfContext(arch, B, T, precision=prec)
ctx.load_state_dict(sd)
ctx.set_schedule(tab.posterior_mean_coef1, tab.posterior_mean_coef2, tab.posterior_log_variance_clipped)
cond = O.det_cond(B, T, tag="x", arch=O.ARCH_MDM_L)
cd = {k: (v.cuda() if hasattr(v, "cuda") else v) for k, v in cond.items()}
ctx.set_cond(cd["text_embedding"], cd["hand_side"], cd["shape"], cd["obj_embedding"], cd["obj_traj"])
ref = ctx.sample_loop(seed=7).clone()
torch.cuda.synchronize()
bad = 0
t0 = time.time()
for i in range(loops):
    out = ctx.sample_loop(seed=7)
    torch.cuda.synchronize()
    if not torch.equal(out, ref):
        bad += 1
        print(f"loop {i}: DIFFERENT bits, max |diff| = {float((out - ref).abs().max()):.3e}", flush=True)
fin = bool(torch.isfinite(ref).all())
print(f"soak {prec} B={B} T={T}: {loops} loops x {N} steps in {time.time() - t0:.1f} s, {bad} differing, finite={fin}, range flag={ctx.status_flags()}", flush=True)
sys.exit(1 if bad or not fin else 0)
