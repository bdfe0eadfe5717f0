"""GPU (library built with -DTAMF_TIMELINE [-DTAMF_TIMELINE_NI=2|4], loaded via TAMF_LIB_OVERRIDE): the stamps of
tools/clip_timeline.py, but of the LAST clip-tile launch of a hipGraph DDPM loop (FFN2 of the last layer, or with
-DTAMF_TIMELINE_NI=4 its FFN1) - the K-tile intervals as they run inside the step, behind the kernels that precede them.
   clip_timeline_insitu.py [prec] [B]"""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oakink2-tamf_amd")]
sys.path.insert(0, os.path.join(ROOT, "tools")); import _ablib  # noqa: E702,F401  (TAMF_LIB_OVERRIDE: A/B builds)
import numpy as np, torch
from oakink2_tamf_amd.hip_backend import TamfContext, lib
from oracle import mdm_oracle as O
prec = sys.argv[1] if len(sys.argv) > 1 else "f16x3"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
T, N = 196, 32
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
for r in range(3):
    ctx.sample_loop(seed=1 + r, out=out)
torch.cuda.synchronize()
buf = np.zeros(512 * 2 * 8 * 4, np.uint64)
rc = lib().tamf_debug_timeline(2, buf.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(buf.nbytes))
assert rc == 0, rc
b = buf.reshape(512, 2, 8, 4)[:256].astype(np.int64)
print(f"{prec} B={B}: last clip-tile launch of the loop")
for h, name, labels in ((0, "X wave 0", ("reads+DMA issue", "MFMAs issued", "barrier wait")), (1, "Y wave 4", ("MFMAs done", "reads done", "barrier wait"))):
    d = np.diff(b[:, h], axis=2)
    ok = (b[:, h, :, 0] > 0).all(axis=1)
    d = d[ok]
    tot = b[ok][:, h, 1:, 0] - b[ok][:, h, :-1, 0]
    print(f"{name}: {ok.sum()} workgroups; interval length median {np.median(tot):.0f} cycles (p10 {np.percentile(tot,10):.0f}, p90 {np.percentile(tot,90):.0f})")
    for i, l in enumerate(labels):
        print(f"    {l:18s} median {np.median(d[:, :, i]):7.0f}  p10 {np.percentile(d[:, :, i],10):7.0f}  p90 {np.percentile(d[:, :, i],90):7.0f}")
