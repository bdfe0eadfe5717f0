"""GPU: ms per DDPM step of the hipGraph loop (B=64, T=196 by default), repeated: loop_time.py [prec] [B] [ddpm_steps] [reps] [tuning] [T] [nograph]
(tuning = tamf_set_gemm_tuning value the graph is captured under, e.g. 0x400fffff; "nograph" = plain launches - what the overlap probe,
selection bit 256 = tuning 0x100fffff of a -DTAMF_BENCH build, needs)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oakink2-tamf_amd")]
sys.path.insert(0, os.path.join(ROOT, "tools")); import _ablib  # noqa: E702,F401  (TAMF_LIB_OVERRIDE: A/B builds)
import torch
from oakink2_tamf_amd.hip_backend import TamfContext
from oracle import mdm_oracle as O
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
N = int(sys.argv[3]) if len(sys.argv) > 3 else 200
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
tune = int(sys.argv[5], 0) if len(sys.argv) > 5 else -1
T = int(sys.argv[6]) if len(sys.argv) > 6 else 196
graph = not (len(sys.argv) > 7 and sys.argv[7] == "nograph")
arch = dict(latent_dim=512, ff_size=2048, num_layers=8, num_heads=4)
sd = O.det_state_dict(O.ARCH_MDM_L, tag="bench/w")
tab = O.make_tables(N, "cosine")
ctx = TamfContext(arch, B, T, precision=prec)
ctx.load_state_dict(sd)
ctx.set_schedule(tab.posterior_mean_coef1, tab.posterior_mean_coef2, tab.posterior_log_variance_clipped)
cond = O.det_cond(B, T, tag="x", arch=O.ARCH_MDM_L)
cd = {k: (v.cuda() if hasattr(v, "cuda") else v) for k, v in cond.items()}
ctx.set_cond(cd["text_embedding"], cd["hand_side"], cd["shape"], cd["obj_embedding"], cd["obj_traj"])
out = torch.empty(B, 99, 1, T, device="cuda")
if tune != -1:
    from oakink2_tamf_amd.hip_backend import lib
    lib().tamf_set_gemm_tuning(tune)
ctx.sample_loop(seed=1, out=out, use_graph=graph); torch.cuda.synchronize()
ts = []
for r in range(reps):
    t = time.perf_counter(); ctx.sample_loop(seed=2 + r, out=out, use_graph=graph); torch.cuda.synchronize(); ts.append((time.perf_counter() - t) / N * 1e3)
print(f"{os.environ.get('TAMF_LIB_OVERRIDE', 'default').split('/')[-1]} {prec} B={B} T={T} tuning {tune:#x} {'graph' if graph else 'plain launches'}: ms/step " + " ".join(f"{t:.3f}" for t in ts), flush=True)
