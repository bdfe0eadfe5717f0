"""GPU experiment: two independent half batches (2 x B/2 clips, two contexts, two streams, two host threads) against one batch of B:
do the kernels of two hipGraph loops overlap on the chip?   two_streams.py [prec] [B] [ddpm_steps] [selection bits for the halves]
Measured (MI355X, f16x3, B = 64): one batch 2.40 ms per step; two halves 2.81 (2 x 1.60 serial = 3.19: they do overlap), 2.66 with
clip tiles allowed at 50 % of the slots (0x400fffff), 2.53 against 2.50 with every kernel of a half restricted to half of the CUs -
the lock-step single batch is not beaten."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oakink2-tamf_amd")]
sys.path.insert(0, os.path.join(ROOT, "tools")); import _ablib  # noqa: E702,F401  (TAMF_LIB_OVERRIDE: A/B builds)
import torch
from oakink2_tamf_amd.hip_backend import TamfContext
from oracle import mdm_oracle as O
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
N = int(sys.argv[3]) if len(sys.argv) > 3 else 200
tune = int(sys.argv[4], 0) if len(sys.argv) > 4 else -1
T = 196
arch = dict(latent_dim=512, ff_size=2048, num_layers=8, num_heads=4)
sd = O.det_state_dict(O.ARCH_MDM_L, tag="bench/w")
tab = O.make_tables(N, "cosine")
def make(b):
    ctx = TamfContext(arch, b, T, precision=prec)
    ctx.load_state_dict(sd)
    ctx.set_schedule(tab.posterior_mean_coef1, tab.posterior_mean_coef2, tab.posterior_log_variance_clipped)
    cond = O.det_cond(b, T, tag="x", arch=O.ARCH_MDM_L)
    cd = {k: (v.cuda() if hasattr(v, "cuda") else v) for k, v in cond.items()}
    ctx.set_cond(cd["text_embedding"], cd["hand_side"], cd["shape"], cd["obj_embedding"], cd["obj_traj"])
    return ctx, torch.empty(b, 99, 1, T, device="cuda")
from oakink2_tamf_amd.hip_backend import lib
whole, out_w = make(B)
whole.sample_loop(seed=1, out=out_w); torch.cuda.synchronize()  # (graph captured with the whole chip's slots)
halves = [make(B // 2) for _ in range(2)]
streams = [torch.cuda.Stream() for _ in range(2)]
def run_whole():
    whole.sample_loop(seed=1, out=out_w); torch.cuda.synchronize()
def run_half(i, seed):
    with torch.cuda.stream(streams[i]):
        halves[i][0].sample_loop(seed=seed, out=halves[i][1])
def run_halves(seed):
    th = [threading.Thread(target=run_half, args=(i, seed + i)) for i in range(2)]
    for t in th: t.start()
    for t in th: t.join()
    torch.cuda.synchronize()
run_whole()
lib().tamf_set_gemm_tuning(tune)  # (the graphs of the halves are captured under this selection)
run_halves(1)
for rep in range(3):
    t = time.perf_counter(); run_whole(); a = (time.perf_counter() - t) / N * 1e3
    t = time.perf_counter(); run_halves(5 + rep); b = (time.perf_counter() - t) / N * 1e3
    t = time.perf_counter(); run_half(0, 9); torch.cuda.synchronize(); c = (time.perf_counter() - t) / N * 1e3
    print(f"{prec}: one batch of {B}: {a:.3f} ms/step | two concurrent halves of {B//2}: {b:.3f} ms/step | one half alone: {c:.3f} ms/step", flush=True)
