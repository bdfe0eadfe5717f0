"""GPU experiment: one B=64 loop vs two concurrent B=32 loops on two streams (same total clips)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oakink2-tamf_amd")]
import torch
from oakink2_tamf_amd.hip_backend import TamfContext
from oracle import mdm_oracle as O
arch = dict(latent_dim=512, ff_size=2048, num_layers=8, num_heads=4)
sd = O.det_state_dict(O.ARCH_MDM_L, tag="bench/w")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
prec = sys.argv[2] if len(sys.argv) > 2 else "bf16x3"
T = 196
tab = O.make_tables(N, "cosine")
def mk(B):
    c = TamfContext(arch, B, T, precision=prec)
    c.load_state_dict(sd); c.set_schedule(tab.posterior_mean_coef1, tab.posterior_mean_coef2, tab.posterior_log_variance_clipped)
    cond = O.det_cond(B, T, tag="x", arch=O.ARCH_MDM_L)
    cd = {k: (v.cuda() if hasattr(v, "cuda") else v) for k, v in cond.items()}
    c.set_cond(cd["text_embedding"], cd["hand_side"], cd["shape"], cd["obj_embedding"], cd["obj_traj"])
    return c
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / reps
c64 = mk(64)
t64 = timeit(lambda: c64.sample_loop(seed=1))
print(f"B=64 single: {t64/N*1e3:.3f} ms/step")
for G in (2, 4):
    ctxs = [mk(64 // G) for _ in range(G)]
    streams = [torch.cuda.Stream() for _ in range(G)]
    def run():
        for c, s in zip(ctxs, streams):
            with torch.cuda.stream(s):
                c.sample_loop(seed=1)
    tg = timeit(run)
    print(f"{G} x B={64//G} concurrent: {tg/N*1e3:.3f} ms/step (per 64 clips)")
